#!/usr/bin/env python3
"""Headline benchmark: segments/sec at po2 = 20 proving the camt53 guest, one process per GPU (BASELINE.json metric, configs[1]).

A step = one pass of the whole path over one batch: every session context of the rank runs `prove(env, elf)` once -- the
hand-assembled hyperfridge guest (tools/guest_camt53.py: RSA-2048 x3, SHA-256, AES-128-CBC, inflate, unzip, camt.053 fields; its
journal is the reference's committed receipt's) is EXECUTED on a host thread, its compact preflight rows are uploaded and expanded
into the trace circuit's DATA group on the device, and its twelve 2^20-row segments are proved in two phases (commit every DATA
group, derive the session challenge, finish every proof) -- executor, witness generation and proofs all inside the timed region.
Sessions are independent receipts, so ranks run their own with no data-path collective ("weak" scaling: BASELINE configs[3]'s
batch of receipts); torch.distributed (RCCL) carries the barrier and the reductions of the timing.  `--sharded-session` proves
ONE session on all ranks instead (configs[2]: segments rank, rank + N, ...; the ranks exchange 28 words per segment between the
phases -- the one collective of the path).

Kept beside it on the same line so that the series r01..r04 stays readable: `synthetic_bench_circuit` (the rounds 1-3 headline:
a resident 256-column synthetic segment per context, r0h_prove_segment alone) and `trace_circuit_resident` (the same protocol
on segments of the real run).  `--circuit bench|small ...` or `--segments S` run that synthetic benchmark as the whole job.

Extra objects on the JSON line:
  roofline      dominant kernel family of a session (largest share of device time): algorithmic HBM bytes / its HIP-event time,
                `traffic` = PMC-measured HBM bytes per launch (profiles/*/pmc_traffic.json, separate rocprofv3 --pmc passes)
  cpu_baseline  the oracle (CPU restatement, OpenMP, as many threads as the cgroup lets run) proving the session's first segment --
                a full 2^20-row trace of the real run -- itself ("port": the risc0 CPU prover cannot be built here); rank 0, N = 1 only
"""
import argparse
import glob
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def device_code_fingerprint(circuit="bench"):
    """SHA-256 over everything that decides which kernels run and what they move: the HIP sources and headers of the library and
    the circuit blob (the eval_check kernels are generated from it).  Stored with a PMC summary when it is collected
    (tools/summarize_pmc.py) and compared here, so that `roofline.traffic` is never quoted from another state of the code."""
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(ROOT, "hyperfridge-r0_amd", "csrc")
    # device code only: the translation units that define kernels and every header they can include (host-only units -- receipts,
    # claims, the verifier, the EBICS pre-processor, the RV32IM executor -- change nothing that runs on the GPU)
    files = []
    for f in sorted(os.listdir(src)):
        path = os.path.join(src, f)
        if f.endswith(".hpp") or (f.endswith(".hip") and b"__global__" in open(path, "rb").read()):
            files.append(path)
    files.append(os.path.join(ROOT, "include", "r0hip_circuit.h"))
    files.append(os.path.join(ROOT, "circuits", circuit + ".r0c"))
    for path in files:
        h.update(os.path.basename(path).encode() + b"\0")
        h.update(open(path, "rb").read())
    return h.hexdigest()


def pmc_traffic(kernel, launches_per_segment=None, circuit="bench"):
    """HBM bytes per launch of `kernel` from the newest committed PMC summary -- or None when that summary was collected on other
    device code than what is running now, or with another launch count per segment for this kernel (a stale figure is worse
    than none: VERDICT r01)."""
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_traffic.json")))
    if not paths:
        return None
    try:
        doc = json.load(open(paths[-1]))
        k = doc["kernels"].get(kernel)
        if not k or doc.get("device_code_sha256") != device_code_fingerprint(circuit):
            return None
        per_segment = doc.get("segments_profiled")
        # (within 2 %: the profiled command commits a CODE group or two more than the accounting session, e.g. for the control roots it verifies against)
        if launches_per_segment is not None and per_segment and abs(k["launches"] / per_segment - launches_per_segment) > 0.02 * launches_per_segment:
            return None
        return k["hbm_bytes_per_launch"]
    except Exception:
        return None


def spawn_ranks(script, argv, n, extra_env=None):
    """`python bench.py --gpus N` typed as such: this process -- which has not touched torch, HIP or the GPU -- starts N fresh
    child processes of the same command line, one rank per GPU, with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, relays
    rank 0's stdout (the JSON line) and returns non-zero if any rank failed.  Nothing is re-exec'd from a process that has
    initialised the GPU.  (The driver's `python -m torch.distributed.run ... bench.py --gpus N` sets WORLD_SIZE itself and
    never reaches this function.)"""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    import signal
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                   R0H_PARENT_PID=str(os.getpid()))  # the rank asks the kernel to end it with this process (die_with_parent)
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()

    # A parent that dies takes exactly its own ranks with it, however it dies: atexit covers an orderly exit; SIGTERM / SIGINT / SIGHUP
    # (a harness time limit, a test's subprocess timeout) are caught and turned into "kill the ranks, exit non-zero"; SIGKILL cannot
    # be caught, so every rank has also armed PR_SET_PDEATHSIG before touching torch or the GPU (die_with_parent).
    def kill_ranks():
        for p in procs:
            if p.poll() is None:
                p.kill()

    import atexit
    atexit.register(kill_ranks)

    def on_signal(signum, _frame):
        kill_ranks()
        os._exit(128 + signum)

    for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        signal.signal(sig, on_signal)
    # a rank that dies before a barrier would leave its peers waiting in it (RCCL: until the watchdog's timeout): as soon as one
    # rank has failed, the others -- exactly the processes started above -- are ended
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):
            time.sleep(2.0)  # let the peers notice by themselves first (their own error message is the better one)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.05)
    codes = [p.wait() for p in procs]
    reader.join(10)
    out = b"".join(c for c in chunks if c)
    sys.stdout.write(out.decode("utf-8", "replace"))
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        sys.stderr.write("bench.py: rank(s) failed: %s\n" % ", ".join("rank %d exit %d" % rc for rc in bad))
        return 1
    return 0


def pin_to_gpu_numa_node(local_rank, min_cores=12):
    """Keep this rank's host threads (one lane thread per in-flight context, transcripts, launches) on the cores of the NUMA node
    its GPU hangs off, when the PCI device's numa_node is readable and names a node whose cores this process may use; otherwise
    leave the affinity as it is.  Eight ranks x eight lane threads on a two-socket host otherwise wander across sockets.
    Returns a short description for the JSON line."""
    try:
        import torch
        pr = torch.cuda.get_device_properties(local_rank)
        bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read())
        if node < 0:
            return "numa_node of %s is -1: affinity left as is" % bdf
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        usable = cpus & os.sched_getaffinity(0)
        if len(usable) < min_cores:  # pinning must never starve the lanes: a node with too few usable cores is not worth it
            return "node %d offers %d usable cores (< %d wanted): affinity left as is" % (node, len(usable), min_cores)
        os.sched_setaffinity(0, usable)  # the calling (main) thread; the lane threads started later inherit it
        return "pinned to NUMA node %d of %s (%d cores)" % (node, bdf, len(usable))
    except Exception as exc:  # noqa: BLE001 -- a sysfs layout we do not know is not an error
        return "not pinned (%s)" % type(exc).__name__


def die_with_parent():
    """In a rank started by spawn_ranks, before any torch / HIP call: have the kernel send SIGKILL to this process when the parent
    dies (prctl PR_SET_PDEATHSIG), and leave at once if the parent is already gone -- otherwise a rank orphaned inside a gloo / RCCL
    barrier keeps its GPU until the collective's own timeout.  Nothing is re-exec'd."""
    parent = os.environ.get("R0H_PARENT_PID")
    if not parent:
        return
    try:
        import ctypes
        import signal
        ctypes.CDLL(None, use_errno=True).prctl(1, int(signal.SIGKILL), 0, 0, 0)  # PR_SET_PDEATHSIG = 1
    except Exception:
        pass
    if os.getppid() != int(parent):  # the parent died between fork and prctl
        os._exit(1)


_RESULT_FD = None


def guard_stdout():
    """The contract is ONE JSON line on stdout.  Libraries underneath write there too -- gloo announces its peers on std::cout,
    RCCL prints its warnings to stdout, build steps chat -- so a process that is going to measure points descriptor 1 at stderr
    for its whole life and keeps the real stdout for the result line alone (emit_result)."""
    global _RESULT_FD
    if _RESULT_FD is None:
        sys.stdout.flush()
        _RESULT_FD = os.dup(1)
        os.dup2(2, 1)


def emit_result(obj):
    data = (json.dumps(obj) + "\n").encode()
    if _RESULT_FD is None:
        sys.stdout.write(data.decode())
        sys.stdout.flush()
    else:
        os.write(_RESULT_FD, data)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--po2", type=int, default=20)
    ap.add_argument("--circuit", default="camt53", help="camt53 (default): the real workload -- sessions of the camt53 guest over the trace circuit; "
                                                       "bench / small / ...: the synthetic benchmark of rounds 1-3 on that circuit blob")
    ap.add_argument("--contexts", type=int, default=0, help="camt53: sessions in flight per GPU (default 2; each has an executor thread and two prover lanes); "
                                                          "synthetic: segments in flight per GPU (default 8; one context + host thread each)")
    ap.add_argument("--sharded-session", action="store_true", help="camt53: all ranks prove ONE session together (segments rank, rank + N, ...; records exchanged "
                                                                    "between the two phases) instead of a session each -- BASELINE configs[2], strong scaling")
    ap.add_argument("--cpu-po2", type=int, default=-1, help="po2 of the CPU-baseline sample: default = --po2 (the headline segment itself, about half a minute "
                                                              "on 16 cores, no scaling); smaller = a bounded sample scaled by rows; 0 disables")
    ap.add_argument("--segments", type=int, default=0, help="BASELINE configs[2]/[3]: prove a fixed batch of this many segments, sharded "
                                                              "round-robin over the ranks (strong scaling); 0 = the default weak-scaling steps")
    ap.add_argument("--seal-dir", default="", help="with --segments: write the seal of every --keep-every-th segment there (seal_<index>.npy) for checking")
    ap.add_argument("--keep-every", type=int, default=8)
    ap.add_argument("--profile-mode", action="store_true", help="for tools/collect_profiles.sh: no prove_elf session and no re-committing comparison, so that "
                                                                 "the process holds warm-up + timed + witgen + accounting segments of ONE configuration")
    ap.add_argument("--no-session", action="store_true", help="(no longer used: the session IS the headline; kept so that older command lines parse)")
    ap.add_argument("--recommit-code", action="store_true", help="commit the CODE group inside every proof (rounds 1-2 behaviour) instead of once per (circuit, po2)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (gloo only for rehearsing ranks on one box)")
    ap.add_argument("--share-device", action="store_true", help="rehearsal: every rank uses device 0")
    ap.add_argument("--rehearse-without-gpu", type=float, default=0.0, metavar="MS",
                    help="TEST ONLY: exercise the multi-process harness (spawn, rendezvous, barrier, reductions, JSON relay) on a box "
                         "without a GPU -- a step sleeps MS milliseconds instead of proving and the line printed says so; implies gloo")
    ap.add_argument("--fail-rank", type=int, default=-1, help="TEST ONLY (with --rehearse-without-gpu): this rank raises inside its step")
    args = ap.parse_args()

    import __graft_entry__ as entry
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no outer launcher: build once here (the ranks then load finished binaries), then one fresh process per GPU
        entry.ensure_built()
        raise SystemExit(spawn_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))

    die_with_parent()
    guard_stdout()
    if args.rehearse_without_gpu > 0:
        return rehearse(args)

    import numpy as np
    import torch

    if int(os.environ.get("LOCAL_RANK", "0")) == 0:
        entry.ensure_built()  # no-op when the binaries travelled with the snapshot
    from hyperfridge_r0_amd import driver  # (importing the package does not load libr0hip.so yet)

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    local_rank = 0 if args.share_device else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    numa = pin_to_gpu_numa_node(local_rank) if args.gpus > 1 and not args.share_device else "single rank: affinity left as is"
    env = driver.DistEnv(backend=args.backend, device=torch.device("cuda", local_rank) if args.backend == "nccl" else None)
    if env.world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, env.world))
    env.barrier()  # local rank 0 may have been building: nobody loads the library before it is finished
    import hyperfridge_r0_amd as r0

    if args.circuit == "camt53" and args.segments:
        args.circuit = "bench"  # a fixed batch of synthetic segments (BASELINE configs[2] on the synthetic circuit: tests/test_gpu_configs.py)
    if args.circuit == "camt53":
        return session_bench(args, env, entry, torch, local_rank, numa)
    line = synthetic_bench(args, env, entry, torch, local_rank, numa)
    if env.rank == 0:
        emit_result(line)
    env.close()


def synthetic_bench(args, env, entry, torch, local_rank, numa, embedded=False):
    """The benchmark of rounds 1-3: a step = one r0h_prove_segment per in-flight context over a resident synthetic segment of
    circuits/<circuit>.r0c (bench: W = (16 CODE, 192 DATA, 48 ACCUM) = 256 columns, ~20k mul + ~22.6k add/sub per point).  Returns the
    JSON line (rank 0) / None; embedded = True: a short run inside the camt53 benchmark, no CPU baseline and no variants."""
    import numpy as np
    import hyperfridge_r0_amd as r0
    from hyperfridge_r0_amd import driver
    po2, n_ctx = args.po2, max(1, args.contexts or 8)
    if args.cpu_po2 < 0:
        args.cpu_po2 = min(po2, 20)
    blob = np.fromfile(entry.circuit_blob_path(args.circuit), dtype=np.uint32)
    co = entry.code_object_path(args.circuit)
    lanes = []
    for k in range(n_ctx):
        hal = r0.Hal(local_rank)
        circuit = hal.load_circuit(blob, co if os.path.exists(co) else None)
        # one resident witness per context (distinct seed per rank and context): inputs are in HBM before timing starts
        code, data, glob_ = hal.witgen(circuit, po2, seed=1000 + env.rank * 16 + k)
        hal.sync()
        lanes.append(dict(hal=hal, circuit=circuit, code=code, data=data, glob=glob_, seal_words=0))
    # CODE depends on (circuit, po2) only (its Merkle root is the control root): committed once per rank, read by every lane's
    # proofs (r0h_code_commit_new / r0h_prove_segment_committed) -- the seals are word for word those of r0h_prove_segment
    code_commit = None if args.recommit_code else lanes[0]["hal"].code_commit(lanes[0]["circuit"], po2, lanes[0]["code"])
    use_commit = [code_commit is not None]

    def prove_on(lane, seed=None, keep_as=None):
        if seed is not None:  # another segment: its witness is generated on the device, into the lane's buffers
            lane["glob"] = lane["hal"].witgen_into(lane["circuit"], po2, seed, lane["code"], lane["data"])
        seal = lane["hal"].prove_segment(lane["circuit"], po2, code_commit if use_commit[0] else lane["code"], lane["data"], lane["glob"])
        lane["seal_words"] = seal.size
        lane["proved"] = lane.get("proved", 0) + 1
        if keep_as is not None:
            lane.setdefault("kept", []).append((keep_as, seal))

    def run_lanes(work):
        """work(lane) on every lane, one host thread each (ctypes releases the GIL inside the library: the contexts' streams
        overlap on the device).  Returns the number of segments actually proved; the first exception of any lane is re-raised
        here, on the main thread, so a failed lane can neither inflate the count nor let the process exit 0."""
        before = sum(ln.get("proved", 0) for ln in lanes)
        errors = []

        def guarded(lane):
            try:
                work(lane)
            except BaseException as exc:  # noqa: BLE001 -- reported below
                errors.append(exc)

        if len(lanes) == 1:
            guarded(lanes[0])
        else:
            ts = [threading.Thread(target=guarded, args=(ln,)) for ln in lanes]
            for t in ts:
                t.start()
            for t in ts:
                t.join()
        if errors:
            raise errors[0]
        return sum(ln.get("proved", 0) for ln in lanes) - before

    def step(_i):
        return run_lanes(prove_on)

    def steps_back_to_back(n):
        """n steps = n segments on every lane; a lane starts its next segment as soon as its previous one is out (no join
        between steps: the lanes are independent provers, exactly as r0h_prove runs them)."""
        def run(lane):
            for _ in range(n):
                prove_on(lane)
        return run_lanes(run)

    def device_sync():
        for ln in lanes:
            ln["hal"].sync()
        torch.cuda.synchronize()

    if args.segments:
        # fixed batch (BASELINE.json configs[2]): rank r owns segments r, r + world, ...; each in-flight context drains its share
        # of them.  Every segment is a different one: its witness (seed 5000 + segment index) is generated on the device inside the
        # timed region, as a deployment would generate it from the preflight trace.
        mine = driver.shard_segments(args.segments, env.world, env.rank)
        shares = [mine[k::n_ctx] for k in range(n_ctx)]
        keep = (lambda s: s if (args.seal_dir and args.keep_every > 0 and s % args.keep_every == 0) else None)

        def drain(_i):
            def run(lane):
                for s in shares[lanes.index(lane)]:
                    prove_on(lane, seed=5000 + s, keep_as=keep(s))
            return run_lanes(run)

        step(-1)
        elapsed, units = driver.run_timed(env, drain, 1, 0, device_sync)
        args.steps, scaling = 1, "strong"
        if args.seal_dir:
            os.makedirs(args.seal_dir, exist_ok=True)
            for ln in lanes:
                for idx, seal in ln.get("kept", []):
                    np.save(os.path.join(args.seal_dir, "seal_%04d.npy" % idx), seal)
        incl = uncached = None
    else:
        elapsed, units = driver.run_timed(env, step, args.steps, args.warmup, device_sync, many_fn=steps_back_to_back if n_ctx > 1 else None)
        scaling = "weak"
        uncached = None
        if code_commit is not None and not args.profile_mode and not embedded:  # the same K steps with CODE committed inside every proof, for comparison.  Not `value`.
            use_commit[0] = False
            el1, un1 = driver.run_timed(env, step, args.steps, 1, device_sync, many_fn=steps_back_to_back if n_ctx > 1 else None)
            uncached = un1 / el1
            use_commit[0] = True
        # the same K steps once more with witness generation inside the timed region (risc0's prove_segment includes it, SURVEY.md
        # 3.4 step 2): every segment gets a fresh synthetic witness, generated on the device into the lane's buffers.  Not `value`.
        counter = [0]

        def steps_with_witgen(n):
            def run(lane):
                k = lanes.index(lane)
                for i in range(n):
                    prove_on(lane, seed=9000 + (counter[0] + i) * 64 + env.rank * 16 + k)
            got = run_lanes(run)
            counter[0] += n
            return got

        incl = None
        if not embedded:
            el2, un2 = driver.run_timed(env, lambda i: steps_with_witgen(1), args.steps, 0, device_sync, many_fn=steps_with_witgen if n_ctx > 1 else None)
            incl = un2 / el2
    # per-kernel accounting: HIP events around every launch, on one context running alone, over as many segments as were
    # timed (outside the timed region, so the events neither perturb `value` nor see another context's kernels)
    lanes[0]["hal"].kernel_timing(True)
    for _ in range(args.steps):
        prove_on(lanes[0])
    kstats = lanes[0]["hal"].kernel_stats()
    phases = lanes[0]["hal"].last_profile()
    lanes[0]["hal"].kernel_timing(False)

    # `prove(env, elf)` end to end, once, beside the headline (rank 0 of a single-GPU run): the hand-assembled guest of
    # tools/guest_camt53.py on the reference's EBICS fixture, executed on a host thread, its compact preflight rows expanded on the device,
    # every segment proved with circuits/trace.r0c -- executor, witness generation and proofs all inside the timed region
    trace_resident = None
    if embedded and env.rank == 0 and po2 == 20:  # the same protocol (witnesses resident, one segment in flight per context) on segments of the real run
        try:
            trace_resident = trace_circuit_resident([ln["hal"] for ln in lanes], entry, args.steps)
        except Exception as exc:  # noqa: BLE001
            trace_resident = {"error": str(exc)[:300]}

    if env.rank == 0:
        steps = max(args.steps, 1)
        value = units / elapsed
        circuit = lanes[0]["circuit"]
        dom = max(kstats.items(), key=lambda kv: kv[1]["total_ms"]) if kstats else None
        roofline = None
        if dom:
            name, st = dom
            launches = max(st["launches"], 1)
            achieved = st["alg_bytes"] / (st["total_ms"] * 1e-3) / 1e9 if st["total_ms"] > 0 else 0.0
            roofline = {"bound": "hbm", "kernel": name, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": pmc_traffic(name, launches / steps, args.circuit),
                        "launches_per_segment": launches / steps, "avg_launch_ms": round(st["total_ms"] / launches, 4),
                        "alg_bytes_per_launch": round(st["alg_bytes"] / launches),
                        "share_of_device_time": round(st["total_ms"] / max(sum(v["total_ms"] for v in kstats.values()), 1e-9), 4),
                        "note": "this kernel is VALU-integer bound (Poseidon2: ~1.36k Montgomery products per permutation, "
                                "the static VALU issue floor of the permutation is in valu_view); its HBM fraction is reported because the metric asks for it: DESIGN.md 6"}
        if roofline and roofline["kernel"] == "hash_rows_kernel":
            # secondary view: the bound this kernel actually sits on -- VALU throughput.  SIMD cycles one wave spends per permutation
            # (1024 SIMDs; the cycles the measured time can hold at the 2.4 GHz maximum clock, so an upper bound of the real count)
            # against the static floor of the compiled permutation: instruction counts x guaranteed minimum issue cycles
            # (tools/p2_issue_floor.py).  frac <= 1 by construction; the rate-table estimate beside it prices the same
            # stream at the rates isolated streams of each instruction sustain.
            rows = 4 << po2
            hashed = [g for k, g in enumerate(circuit.group_size) if not (code_commit is not None and k == 1)]  # CODE is hashed once, outside
            perms = rows * sum(-(-g // 16) for g in hashed) + rows  # the groups hashed per segment + CHECK (16 columns)
            d = 1 << po2
            while d > 256:  # FRI rounds: 4d/16 rows of 64 columns
                perms += (4 * d // 16) * 4
                d //= 16
            st = kstats["hash_rows_kernel"]
            cyc = st["total_ms"] * 1e-3 * 2.4e9 * 1024 / (perms * steps / 64.0)
            view = {"permutations_per_segment": perms, "G_permutations_per_s": round(perms * steps / (st["total_ms"] * 1e-3) / 1e9, 3),
                    "simd_cycles_per_wave_permutation_at_2.4GHz": round(cyc, 1), "issue_floor_simd_cycles": None, "frac": None}
            floors = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "p2_issue_floor.json")))
            if floors:
                fl = json.load(open(floors[-1]))
                view["issue_floor_simd_cycles"] = fl["issue_floor_simd_cycles_per_wave_permutation"]
                view["rate_table_estimate_simd_cycles"] = fl.get("rate_table_estimate_simd_cycles_per_wave_permutation")
                view["valu_instructions_per_permutation"] = fl["valu_instructions_per_permutation"]
                view["multiply_instructions_per_permutation"] = sum(v for k, v in fl["by_class"].items() if k != "full_rate")
                view["frac"] = round(view["issue_floor_simd_cycles"] / cyc, 4)
            roofline["valu_view"] = view
        cols = sum(circuit.group_size)
        seg_bytes = (68 * cols + 3132) * (1 << po2)  # Bytes(C) = 68 MiB*C + 3132 MiB at 2^20 rows
        line = {
            "metric": "segments/sec (po2=%d) through r0h_prove_segment on a resident synthetic segment of %s.r0c (the rounds 1-3 headline)" % (po2, args.circuit),
            "value": round(value, 4), "unit": "segments/s", "n_gpus": env.world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / steps * 1e3, 3), "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "value_with_witgen_in_timed_region": round(value, 4) if incl is None else round(incl, 4),
            "value_with_code_committed_in_every_proof": None if uncached is None else round(uncached, 4),
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": ("configs[2] shape" if args.segments else "configs[1] shape") + ": 2^%d-row segments, synthetic circuit %s.r0c W=(%d code,%d data,%d accum), %d segment(s) "
                                   "in flight per GPU, %s; no bundled camt53 trace exists (needs the risc0 executor)" % (
                                       po2, args.circuit, circuit.group_size[1], circuit.group_size[2], circuit.group_size[0], n_ctx,
                                       ("fixed batch of %d distinct segments, witness generated on the device inside the timed region" % args.segments if args.segments
                                        else "witness resident in HBM") + ("; CODE committed in every proof" if code_commit is None else
                                                                            "; CODE commitment cached per (circuit, po2): committed once per rank before timing, read by every proof")),
                       "po2": po2, "columns": cols, "taps": circuit.n_taps, "seal_words": int(lanes[0]["seal_words"]),
                       "segments_per_step_per_gpu": n_ctx, "fixed_batch_segments": args.segments or None,
                       "parallelism": "segment-parallel x%d" % env.world, "host_threads": "rank 0: " + numa},
            "roofline": roofline,
            "cpu_baseline": cpu_baseline(blob, args.cpu_po2, po2) if (args.cpu_po2 and env.world == 1 and not embedded) else None,
            "segment_hbm_model": {"alg_bytes_per_segment": seg_bytes, "achieved_GBs_per_gpu": round(seg_bytes * value / env.world / 1e9, 2),
                                  "frac_of_8TBs": round(seg_bytes * value / env.world / 1e9 / HBM_PEAK_GBS, 5)},
            "trace_circuit_resident": trace_resident,
            "phases_ms_last_segment": {n: round(ms, 3) for n, ms in phases},
            "kernels_ms_per_segment": {k: round(v["total_ms"] / steps, 3) for k, v in sorted(kstats.items(), key=lambda kv: -kv[1]["total_ms"])},
            # algorithmic bytes / device time of each kernel family (per-kernel achieved GB/s)
            "kernels_alg_GBs": {k: round(v["alg_bytes"] / (v["total_ms"] * 1e-3) / 1e9, 1) for k, v in sorted(kstats.items(), key=lambda kv: -kv[1]["total_ms"])
                                if v["total_ms"] > 0 and v.get("alg_bytes")},
        }
    else:
        line = None
    if code_commit is not None:
        code_commit.free()
    for ln in lanes:
        for key in ("code", "data", "circuit"):
            ln[key].free()
        ln["hal"].close()
    return line


def session_bench(args, env, entry, torch, local_rank, numa):
    """The headline: sessions of the camt53 guest, end to end (module docstring)."""
    import numpy as np
    import hyperfridge_r0_amd as r0
    from hyperfridge_r0_amd import driver
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import guest_camt53
    po2 = args.po2
    image, stream, what = guest_camt53.elf_and_input(form=1)
    want_journal = bytes(json.load(open(os.path.join(ROOT, "tests", "golden", "reference_receipt_6bb95807_latest.json")))["journal"]["bytes"])
    blob = np.fromfile(entry.circuit_blob_path("trace"), dtype=np.uint32)
    n_ses = 1 if args.sharded_session else max(1, args.contexts or 2)
    if not args.contexts and not args.sharded_session:
        # two sessions keep about five host threads busy per rank (two executors, four lanes waiting on the device): on a node whose
        # cgroup gives the job fewer than that per rank, one session per GPU loses a tenth of the rate instead of far more
        share = host_threads()[0] / max(1, env.world)
        if share < 5:
            n_ses = 1
    lanes = []
    image_blob = np.fromfile(entry.circuit_blob_path("image"), dtype=np.uint32)
    for k in range(n_ses):
        hal = r0.Hal(local_rank)
        lanes.append(dict(hal=hal, gc=hal.load_circuit(blob, entry.code_object_path("trace")), segments=0, walls=[], receipt=None))
        if not args.sharded_session:  # every receipt carries its image proof (inside the timed region): `receipt.verify(image_id)` then needs no ELF
            lanes[-1]["ic"] = hal.load_circuit(image_blob, entry.code_object_path("image"))
            hal.set_image_circuit(lanes[-1]["ic"])

    def one_session(lane):
        t0 = time.perf_counter()
        if args.sharded_session:
            receipt, image_id, cycles = driver.prove_elf_sharded(env, lane["hal"], lane["gc"], image, stream, segment_po2=po2)
            n = lane["hal"].last_session_stats()["segments"]  # this rank's share
        else:
            receipt, image_id, cycles = lane["hal"].prove_elf(lane["gc"], image, stream, segment_po2=po2)
            n = lane["hal"].last_session_stats()["segments"]
        lane["walls"].append(time.perf_counter() - t0)
        lane["receipt"], lane["cycles"], lane["segments"] = receipt, cycles, lane["segments"] + n
        return n

    def run_lanes(n_each):
        """every session context runs n_each sessions back to back on its own host thread (a context starts its next session as
        soon as its previous receipt is out); returns the segments proved; a failed lane re-raises here"""
        before = sum(ln["segments"] for ln in lanes)
        errors = []

        def guarded(lane):
            try:
                for _ in range(n_each):
                    one_session(lane)
            except BaseException as exc:  # noqa: BLE001 -- reported below
                errors.append(exc)

        if len(lanes) == 1:
            guarded(lanes[0])
        else:
            ts = [threading.Thread(target=guarded, args=(ln,)) for ln in lanes]
            for t in ts:
                t.start()
            for t in ts:
                t.join()
        if errors:
            raise errors[0]
        return sum(ln["segments"] for ln in lanes) - before

    def device_sync():
        for ln in lanes:
            ln["hal"].sync()
        torch.cuda.synchronize()

    elapsed, units = driver.run_timed(env, lambda i: run_lanes(1), args.steps, args.warmup, device_sync, many_fn=run_lanes if n_ses > 1 else None)
    timed_walls = sorted(w for ln in lanes for w in ln["walls"][-args.steps:])
    stats = lanes[0]["hal"].last_session_stats()
    line = None
    if env.rank == 0:
        # ---- outside the timed region: every lane's last receipt verified the way the reference's verifier would, with the ELF
        receipt = lanes[0]["receipt"]
        seals = receipt.seals()
        roots = {}
        for _, seal in seals:
            size = r0.verify_seal(blob, seal)[2]
            if size not in roots:
                cc = lanes[0]["hal"].code_commit(lanes[0]["gc"], size)
                roots[size] = cc.root()
                cc.free()
        verified = all(ln["receipt"].verify(blob, roots, None, elf=image)[:2] == (0, "ok") for ln in lanes if ln["receipt"] is not None)
        # ... and the way the reference's verifier calls it: the image id's 32 bytes, no ELF (the receipt's image proof stands for the image)
        image_id = r0.compute_image_id(image)
        by_id = None if args.sharded_session else all(ln["receipt"].verify_image(blob, roots, image_blob, image_id)[:2] == (0, "ok") for ln in lanes if ln["receipt"] is not None)
        n_seg = len(seals)
        # ---- per-kernel accounting: one session alone on one context with ONE prover lane (so that every launch is on the context
        # whose HIP events are read), outside the timed region
        os.environ["R0H_SESSION_LANES"] = "1"
        try:
            lanes[0]["hal"].kernel_timing(True)
            lanes[0]["hal"].prove_elf(lanes[0]["gc"], image, stream, segment_po2=po2)
            kstats = lanes[0]["hal"].kernel_stats()
            phases = lanes[0]["hal"].last_profile()
            lanes[0]["hal"].kernel_timing(False)
        finally:
            del os.environ["R0H_SESSION_LANES"]
        gs = lanes[0]["gc"].group_size
        roofline = None
        if kstats:
            name, st = max(kstats.items(), key=lambda kv: kv[1]["total_ms"])
            launches = max(st["launches"], 1)
            achieved = st["alg_bytes"] / (st["total_ms"] * 1e-3) / 1e9 if st["total_ms"] > 0 else 0.0
            roofline = {"bound": "hbm", "kernel": name, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                        "traffic": pmc_traffic(name, launches / n_seg, "trace"), "launches_per_segment": round(launches / n_seg, 3), "avg_launch_ms": round(st["total_ms"] / launches, 4),
                        "alg_bytes_per_launch": round(st["alg_bytes"] / launches),
                        "share_of_device_time": round(st["total_ms"] / max(sum(v["total_ms"] for v in kstats.values()), 1e-9), 4),
                        "note": "measured over one whole session (%d segments) on one context with one prover lane; this kernel is VALU-integer bound "
                                "(Poseidon2: ~1.36k Montgomery products per permutation); its HBM fraction is reported because the metric asks for it: DESIGN.md 6" % n_seg}
        # ---- the series of rounds 1-3, kept readable: the synthetic 256-column circuit and the resident trace segments (short runs)
        synthetic = None
        if env.world == 1 and not args.profile_mode and po2 == 20:
            for ln in lanes:  # (the sessions' pooled buffers go back to the device first)
                ln["hal"].sync()
            import copy
            sargs = copy.copy(args)
            sargs.circuit, sargs.contexts, sargs.steps, sargs.warmup, sargs.cpu_po2, sargs.segments = "bench", 8, 3, 1, 0, 0
            try:
                sl = synthetic_bench(sargs, env, entry, torch, local_rank, numa, embedded=True)
                synthetic = {"value": sl["value"], "unit": sl["unit"], "ms_per_step": sl["ms_per_step"], "steps": 3, "segments_in_flight": 8, "workload": sl["config"]["workload"],
                             "roofline": sl["roofline"], "trace_circuit_resident": sl["trace_circuit_resident"]}
            except Exception as exc:  # noqa: BLE001 -- the headline does not depend on it
                synthetic = {"error": str(exc)[:300]}
        cols = sum(gs)
        seg_bytes = (68 * cols + 3132) * (1 << po2)  # Bytes(C) = 68 MiB*C + 3132 MiB at 2^20 rows
        value = units / elapsed
        steps = max(args.steps, 1)
        line = {
            "metric": json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"],
            "value": round(value, 4), "unit": "segments/s", "n_gpus": env.world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / steps * 1e3, 3), "higher_is_better": True, "scaling": "strong" if args.sharded_session else "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": "configs[1]: the camt53 guest at po2 = 20 -- prove(env, elf) end to end: %s; executed on a host thread (%d cycles, %d segments of at most 2^20 rows), rows "
                                   "expanded on the device, proved with circuits/trace.r0c W=(%d accum, %d code, %d data) in two phases under the session challenge, and the image proof of circuits/image.r0c attached (inside the timed region); %s; every receipt "
                                   "verified with the ELF, and with the image id alone, after the timed region.  data is 'synthetic' in the contract's sense only: the guest is hand-assembled (the reference ships no ELF) and its "
                                   "input is the reference's own EBICS fixture" % (what, lanes[0]["cycles"], n_seg, gs[0], gs[1], gs[2],
                                   "ONE session sharded over all ranks (records all-reduced between the phases)" if args.sharded_session else "%d session(s) in flight per GPU, each with an executor thread and two prover lanes; a step = every session context proves one session" % n_ses),
                       "po2": po2, "columns": cols, "segments_per_session": n_seg, "sessions_per_step_per_gpu": n_ses, "seal_words": int(seals[0][1].size),
                       "parallelism": ("one session over %d GPUs" if args.sharded_session else "independent sessions x%d GPUs") % env.world, "host_threads": "rank 0: " + numa},
            "receipts_verified_with_the_elf": bool(verified),
            "receipts_verified_with_the_image_id_alone": by_id,
            "journal_is_the_reference_receipt_fixtures": receipt.journal == want_journal,
            "journal": r0.journal_commitment(receipt.journal).decode()[:80] + " ...",
            "per_session_wall_s": {"median": round(timed_walls[len(timed_walls) // 2], 4), "min": round(timed_walls[0], 4), "max": round(timed_walls[-1], 4), "sessions_timed": len(timed_walls)},
            "session_stages": {"executor_MHz_with_trace_kept": round(stats["cycles"] / max(stats["executor_s"], 1e-9) / 1e6, 1), "executor_host_ms_per_segment": round(1e3 * stats["executor_s"] / max(stats["segments"], 1), 2),
                               "witgen_ms_per_segment": round(stats["witgen_ms"] / max(stats["segments"], 1), 2), "prove_ms_per_segment_on_a_lane": round(stats["prove_ms"] / max(stats["segments"], 1), 2)},
            "roofline": roofline,
            "cpu_baseline": cpu_baseline_trace(blob, image, stream, po2) if (args.cpu_po2 and env.world == 1) else None,
            "segment_hbm_model": {"alg_bytes_per_segment": seg_bytes, "achieved_GBs_per_gpu": round(seg_bytes * value / env.world / 1e9, 2),
                                  "frac_of_8TBs": round(seg_bytes * value / env.world / 1e9 / HBM_PEAK_GBS, 5)},
            "synthetic_bench_circuit": synthetic,
            "phases_ms_last_segment": {n: round(ms, 3) for n, ms in phases},
            "kernels_ms_per_segment": {k: round(v["total_ms"] / n_seg, 3) for k, v in sorted(kstats.items(), key=lambda kv: -kv[1]["total_ms"])},
            "kernels_alg_GBs": {k: round(v["alg_bytes"] / (v["total_ms"] * 1e-3) / 1e9, 1) for k, v in sorted(kstats.items(), key=lambda kv: -kv[1]["total_ms"])
                                if v["total_ms"] > 0 and v.get("alg_bytes")},
        }
        emit_result(line)
    for ln in lanes:
        ln["receipt"] = None
        if ln.get("ic") is not None:
            ln["hal"].set_image_circuit(None)
            ln["ic"].free()
        ln["gc"].free()
        ln["hal"].close()
    env.close()
    return 0


def trace_circuit_resident(hals, entry, steps):
    """The synthetic benchmark's protocol on the real circuit: the first len(hals) segments of the camt53 guest's run (2^20 rows each), their
    witnesses expanded once and resident in HBM, one in flight per context, each proved `steps` times back to back with circuits/trace.r0c
    under a fixed challenge.  No executor, no witness generation and no session phases in the timed region -- what is left is the device's
    rate on 138 + 40 columns of real trace."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import threading
    import numpy as np
    import guest_camt53
    import hyperfridge_r0_amd as r0
    image, stream, _ = guest_camt53.elf_and_input(form=1)
    vm = r0.Vm()
    vm.load_elf(image)
    vm.set_input(stream)
    traces = []
    while len(traces) < len(hals):
        finished, _, _ = vm.run_segment(segment_po2=20, keep_trace=True, boundary_rows=True)
        k = len(vm.segments()) - 1
        if finished:
            break  # the last segment is a short one: left out
        rows, bounds = vm.preflight_arrays(k)
        traces.append((rows.copy(), bounds.copy(), vm.claims()[k].globals(), k + 1))
        vm.release_trace(k)
    blob = np.fromfile(entry.circuit_blob_path("trace"), dtype=np.uint32)
    lanes, cc, code = [], None, None
    try:
        for hal, (rows, bounds, claim, number) in zip(hals, traces):
            gc = hal.load_circuit(blob, entry.code_object_path("trace"))
            lanes.append(dict(hal=hal, gc=gc))
            if cc is None:
                code, synthetic, _ = hal.witgen(gc, 20, 0)
                synthetic.free()
                cc = hal.code_commit(gc, 20, code)
            data, glob = hal.trace_witgen(rows, bounds, 20, claim_globals=claim, number=number, closing=False, circuit=gc)
            glob[r0.TRACE_GAMMA:r0.TRACE_GAMMA + 16] = np.arange(3, 19, dtype=np.uint32)  # a fixed challenge: these seals stand outside any session
            lanes[-1]["data"], lanes[-1]["glob"] = data, hal.logup_totals(gc, 20, code, data, glob)
        errors = []

        def run(lane, n):
            try:
                for _ in range(n):
                    lane["words"] = lane["hal"].prove_segment(lane["gc"], 20, cc, lane["data"], lane["glob"]).size
            except BaseException as exc:  # noqa: BLE001 -- re-raised below
                errors.append(exc)

        def all_lanes(n):
            ts = [threading.Thread(target=run, args=(ln, n)) for ln in lanes]
            t0 = time.perf_counter()
            for t in ts:
                t.start()
            for t in ts:
                t.join()
            for ln in lanes:
                ln["hal"].sync()
            if errors:
                raise errors[0]
            return time.perf_counter() - t0

        all_lanes(1)
        n = max(1, steps)
        wall = all_lanes(n)
        return {"value": round(len(lanes) * n / wall, 3), "unit": "segments/s", "segments_in_flight": len(lanes), "steps": n, "ms_per_step": round(1e3 * wall / n, 2),
                "circuit": "trace.r0c W=(%d accum, %d code, %d data)" % tuple(lanes[0]["gc"].group_size), "seal_words": int(lanes[0]["words"]),
                "workload": "the first %d segments (2^20 rows each) of the camt53 guest's run, witnesses resident in HBM, CODE committed once, a fixed challenge" % len(lanes)}
    finally:
        for ln in lanes:
            if "data" in ln:
                ln["data"].free()
            ln["gc"].free()
        if cc is not None:
            cc.free()
        if code is not None:
            code.free()


def host_threads():
    """(threads to use, cores in the affinity mask, cgroup CPU quota): as many threads as the cgroup lets run at once -- a box may expose
    256 cores in the affinity mask to a job whose share is 16 (round 3 calibrated over 256 threads for 14 s to find that out)"""
    avail = os.cpu_count() or 1
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        pass
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, int(round(int(q) / int(per))))
    except Exception:
        pass
    return min(avail, quota or avail), avail, quota


def cpu_baseline_trace(blob, image, stream, po2):
    """The oracle (CPU restatement, OpenMP) proving the FIRST segment of the camt53 session -- a full 2^20-row trace of the real run,
    the unit `value` counts -- from the host reference witness, under a fixed challenge.  Only the proof is timed (the executor and the
    witness expansion are host work on both sides)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import hyperfridge_r0_amd as r0
    import orc_binding
    orc = orc_binding.load()
    cores, avail, quota = host_threads()
    orc.L.orc_set_threads(cores)
    oc = orc.circuit(blob)
    vm = r0.Vm()
    vm.load_elf(image)
    vm.set_input(stream)
    vm.run_segment(segment_po2=po2, keep_trace=True, boundary_rows=True)
    seg = vm.segments()[0]
    size = max(r0.TRACE_MIN_PO2, int(np.ceil(np.log2(seg.user_cycles + seg.boundary_rows))))
    data, glob = vm.trace_witness(0, size, claim_globals=vm.claims()[0].globals())
    vm.close()
    code = oc.witgen(size, 0)[0]
    glob[r0.TRACE_GAMMA:r0.TRACE_GAMMA + 16] = np.arange(3, 19, dtype=np.uint32)
    glob = oc.logup_totals(size, code, data, glob)
    t0 = time.perf_counter()
    seal = oc.prove(size, code, data, glob)
    dt = time.perf_counter() - t0
    ok = oc.verify(seal, code_root=oc.code_root(code, size))[0] == 0
    return {"value": round(1.0 / dt, 6), "unit": "segments/s", "cores": cores, "kind": "port", "cores_in_affinity_mask": avail, "cgroup_cpu_quota": quota,
            "sample": "oracle/liborc.so (C, OpenMP) proved the first segment of the camt53 session itself -- %d cycles + %d boundary rows in a 2^%d-row trace of circuits/trace.r0c -- in "
                      "%.2f s on %d threads (%d cores in the affinity mask, cgroup quota %s); seal %d words, accepted by the oracle's verifier: %s.  No scaling.  The risc0 CPU prover "
                      "cannot be built here (Rust)." % (seg.user_cycles, seg.boundary_rows, size, dt, cores, avail, quota, seal.size, ok)}


def rehearse(args):
    """The harness around the proving step, without the proving step (no GPU needed): what tests/test_distributed.py drives
    through `python bench.py --gpus 2 --rehearse-without-gpu MS`.  The printed line is NOT a measurement and is labelled so."""
    from hyperfridge_r0_amd import driver
    env = driver.DistEnv(backend="gloo")
    if env.world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, env.world))
    n_ctx = max(1, args.contexts)

    def step(i):
        if env.rank == args.fail_rank and i >= 0:
            raise RuntimeError("rehearsal: rank %d fails on purpose" % env.rank)
        time.sleep(args.rehearse_without_gpu * 1e-3)
        return n_ctx

    elapsed, units = driver.run_timed(env, step, args.steps, args.warmup)
    if env.rank == 0:
        emit_result({"metric": "REHEARSAL of the multi-process harness -- no proving, not a measurement", "value": round(units / elapsed, 4),
                     "unit": "sleeps/s", "n_gpus": env.world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 3),
                     "data": "none", "scaling": "weak"})
    env.close()
    return 0


def cpu_baseline(blob, cpu_po2, po2):
    """Oracle (CPU restatement, OpenMP) proving the same circuit on every host core this process may run on.  At the default
    (cpu_po2 == po2 == 20) it proves the headline-size segment itself -- no scaling -- and a 2^18-row segment (BASELINE.json
    configs[0]) as a second sample; a smaller --cpu-po2 gives the bounded sample scaled by rows, labelled as such."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc_binding
    orc = orc_binding.load()
    cores, avail, quota = host_threads()  # as many threads as the cgroup lets run at once (no calibration over the whole affinity mask)
    calib = {}
    oc = orc.circuit(blob)
    orc.L.orc_set_threads(cores)

    def timed(p):
        code, data, glob_ = oc.witgen(p, seed=1000)
        t0 = time.perf_counter()
        seal = oc.prove(p, code, data, glob_)
        return time.perf_counter() - t0, seal.size

    dt, words = timed(cpu_po2)
    scale = 1 << (po2 - cpu_po2)
    out = {"value": round(1.0 / (dt * scale), 6), "unit": "segments/s", "cores": cores, "kind": "port",
           "cores_in_affinity_mask": avail, "cgroup_cpu_quota": quota, "thread_calibration_s_at_po2_12": calib or None,
           "sample": "oracle/liborc.so (C, OpenMP) proved one 2^%d-row segment of the same circuit in %.2f s on %d threads (%d cores in the affinity "
                     "mask, cgroup quota %s; threads = what the cgroup lets run at once%s)%s; seal %d words.  The risc0 CPU prover cannot "
                     "be built here (Rust)." % (cpu_po2, dt, cores, avail, quota, "",
                                                "" if scale == 1 else "; scaled x%d by rows to 2^%d (favours the CPU: drops the log factor)" % (scale, po2), words)}
    if scale == 1 and po2 > 18:
        dt18, words18 = timed(18)
        out["sample_po2_18"] = {"seconds": round(dt18, 3), "segments_per_s_at_po2_18": round(1.0 / dt18, 5), "seal_words": words18,
                                "note": "BASELINE.json configs[0] size (one 2^18-row segment), same circuit, same threads"}
    return out


if __name__ == "__main__":
    sys.exit(main())
