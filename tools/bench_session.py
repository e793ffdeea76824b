#!/usr/bin/env python3
"""`prove(env, elf)` end to end with the trace circuit, timed: the executor (host thread), the device expansion of the compact
preflight rows (r0h_trace_witgen, upload included) and the proof of every segment are all inside the timed region -- one JSON line.

The guest is hand-assembled (the reference ships no ELF): by default the memory-traffic loop of tests/test_rv32im.py (_guest),
sized by --cycles; with --guest rsa the program of tools/guest_rsa.py (SHA-256 + three RSA-2048 public-key operations on the reference's
own inputs); with --guest camt53 the whole pipeline of tools/guest_camt53.py (RSA, SHA-256, AES-128-CBC, inflate, unzip, camt.053 fields),
whose journal is the reference's committed receipt's.  What is reported: segments per second of the whole pipeline, the executor's rate,
host milliseconds per segment (executor thread; it overlaps the device), device-side milliseconds per segment for witness
generation and proof.  The receipt is verified against the image id before the line is printed.
usage: python tools/bench_session.py [--po2 20] [--cycles 8000000] [--guest loop|rsa|camt53] [--repeat 2]
Several GPUs: launch it under `python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 ... tools/bench_session.py ...`:
every rank executes the guest and commits segments rank, rank + N, ...; the ranks exchange their segments' records (28 words each: one
all-reduce -- the session challenge depends on every segment), finish their proofs, and rank 0 collects the receipts
(driver.prove_elf_sharded), merges and verifies them with the ELF and prints the line.
--backend gloo --share-device rehearses that on one GPU."""
import argparse
import json
import os
import struct
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def elf_of(words, base, data=b"", data_addr=0):
    code = struct.pack("<%dI" % len(words), *words)
    n_ph = 2 if data else 1
    ehdr = b"\x7fELF" + bytes([1, 1, 1, 0]) + bytes(8) + struct.pack("<HHIIIIIHHHHHH", 2, 243, 1, base, 52, 0, 0, 52, 32, n_ph, 0, 0, 0)
    o1 = 52 + 32 * n_ph
    ph = struct.pack("<IIIIIIII", 1, o1, base, base, len(code), len(code), 5, 4)
    if data:
        data = data + bytes(-len(data) % 4)
        ph += struct.pack("<IIIIIIII", 1, o1 + len(code), data_addr, data_addr, len(data), len(data), 6, 4)
    return ehdr + ph + code + data


def commitment(r0, journal):
    try:
        return r0.journal_commitment(journal).decode("utf-8", "replace")[:160]
    except r0.R0HipError:
        return None  # the loop guest commits two plain words, not a serde-framed JSON


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--po2", type=int, default=20)
    ap.add_argument("--cycles", type=int, default=8_000_000)
    ap.add_argument("--guest", default="loop")
    ap.add_argument("--repeat", type=int, default=2, help="runs; the last one is reported (the first pays for code objects, pools, CODE commitments)")
    ap.add_argument("--oracle-check", type=int, default=1, help="verify this many seals with the CPU oracle's verifier as well")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend when launched with WORLD_SIZE > 1 (nccl = RCCL)")
    ap.add_argument("--share-device", action="store_true", help="all ranks on GPU 0 (rehearsal on a one-GPU box)")
    ap.add_argument("--sessions", type=int, default=1, help="sessions in flight on this GPU, each on a context of its own (single rank only)")
    ap.add_argument("--resident-limit-gb", type=float, default=-1.0, help="r0h_ctx_set_session_resident_limit per session context (default: the library's, an eighth of the device)")
    args = ap.parse_args()
    world, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    import __graft_entry__ as entry
    env = None
    if world > 1:  # the order bench.py keeps: torch and the process group first, local rank 0 builds, the library is loaded after the barrier
        import torch
        from hyperfridge_r0_amd import driver
        if local == 0:
            entry.ensure_built()
        index = 0 if args.share_device else local
        torch.cuda.set_device(index)
        env = driver.DistEnv(backend=args.backend, device=torch.device("cuda", index))
        env.barrier()
    else:
        entry.ensure_built()
    import hyperfridge_r0_amd as r0
    if args.guest in ("rsa", "camt53"):
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        mod = __import__("guest_" + args.guest)
        elf, stream, what = mod.elf_and_input()
    else:
        from test_rv32im import ADDI, A0, A1, A7, B, ECALL, I, LI, R, S, S0, T0, T1, T2, flat
        buf, scratch, n_loop = r0.JOURNAL_BASE, 0x20000, max(1, (args.cycles - 40) // 7)  # (COMMIT names words of the journal window)
        # reads two input words, then n_loop times: store the counter at scratch + ((t1 & 0x7fc) << 4) (a 32 KiB window, 32 pages), count
        prog = flat(LI(A0, buf), ADDI(A1, 0, 2), ADDI(A7, 0, 1), ECALL, LI(S0, scratch), LI(T2, n_loop), ADDI(T1, 0, 0),
                    I(0x7FC, T1, 7, T0, 0x13), I(4, T0, 1, T0, 0x13), R(0, S0, T0, 0, T0), S(0, T1, T0, 2), ADDI(T1, T1, 4), ADDI(T2, T2, -1), B(-24, 0, T2, 1),
                    LI(A0, buf), I(0, A0, 2, T0, 0x03), R(0, T1, T0, 0, T0), S(0, T0, A0, 2), ADDI(A1, 0, 2), ADDI(A7, 0, 2), ECALL,
                    ADDI(A0, 0, 0), ADDI(A7, 0, 0), ECALL)
        elf, stream = elf_of(prog, 0x400), [7, 0x01020304]
        what = "store loop over a 32 KiB window (7 instructions per iteration, one store), about %d cycles" % args.cycles
    blob = np.fromfile(entry.circuit_blob_path("trace"), dtype=np.uint32)
    hal = r0.Hal(0 if (env is None or args.share_device) else local)
    gc = hal.load_circuit(blob, entry.code_object_path("trace"))
    if args.resident_limit_gb >= 0:
        hal.set_session_resident_limit(int(args.resident_limit_gb * (1 << 30)))
    others = []  # further sessions in flight beside this one: a context and a loaded circuit each
    if env is None and args.sessions > 1:
        import threading
        for _ in range(args.sessions - 1):
            h2 = r0.Hal(0)
            if args.resident_limit_gb >= 0:
                h2.set_session_resident_limit(int(args.resident_limit_gb * (1 << 30)))
            others.append((h2, h2.load_circuit(blob, entry.code_object_path("trace"))))
    peak_used = 0
    for _ in range(max(1, args.repeat)):
        if env is not None:
            env.barrier()
        t0 = time.perf_counter()
        if env is None and others:
            import torch  # (only to read the device's free memory while the sessions run)
            results, stop = [None] * len(others), threading.Event()

            def side(i):
                results[i] = others[i][0].prove_elf(others[i][1], elf, stream, segment_po2=args.po2)

            def watch():
                nonlocal peak_used
                while not stop.wait(0.02):
                    free_b, total_b = torch.cuda.mem_get_info(0)
                    peak_used = max(peak_used, total_b - free_b)

            threads = [threading.Thread(target=side, args=(i,)) for i in range(len(others))] + [threading.Thread(target=watch)]
            [t.start() for t in threads]
            receipt, image_id, cycles = hal.prove_elf(gc, elf, stream, segment_po2=args.po2)
            [t.join() for t in threads[:-1]]
            stop.set()
            threads[-1].join()
            assert all(r is not None and r[0].journal == receipt.journal for r in results)
        elif env is None:
            receipt, image_id, cycles = hal.prove_elf(gc, elf, stream, segment_po2=args.po2)
        else:
            receipt, image_id, cycles = driver.prove_elf_sharded(env, hal, gc, elf, stream, segment_po2=args.po2)
            env.barrier()
        wall = time.perf_counter() - t0
        st = hal.last_session_stats()
    if env is not None and env.rank != 0:
        gc.free()
        hal.close()
        env.close()
        return
    seals = receipt.seals()
    roots = {}
    for _, seal in seals:
        size = r0.verify_seal(blob, seal)[2]
        if size not in roots:
            cc = hal.code_commit(gc, size)
            roots[size] = cc.root()
            cc.free()
    verdict = receipt.verify(blob, roots, None, elf=elf)  # with the ELF: image id, seals, claims, the session's challenge and balance
    assert verdict[:2] == (0, "ok"), verdict
    if args.oracle_check:
        import orc_binding
        oc = orc_binding.load().circuit(blob)
        for _, seal in seals[:args.oracle_check]:
            assert oc.verify(seal, code_root=roots[r0.verify_seal(blob, seal)[2]]) == (0, "ok")
    n = len(seals) * max(1, args.sessions if env is None else 1)
    line = {"sessions_in_flight": args.sessions if env is None else 1, "lean_segments_of_the_reported_session": st["lean_segments"],
            "peak_device_memory_used_GiB": round(peak_used / (1 << 30), 1) if peak_used else None, "metric": "segments/s of prove(env, elf) with the trace circuit: executor + device witgen + proof, all inside the timed region",
            "value": round(n / wall, 4), "unit": "segments/s", "n_gpus": world, "sharding": None if env is None else "segments rank, rank + %d, ... per rank (%s%s); receipts merged on rank 0" % (world, args.backend, ", all ranks on one GPU" if args.share_device else ""), "segment_po2": args.po2, "segments": len(seals), "cycles": cycles,
            "wall_s": round(wall, 4), "guest": what,
            "executor": {"host_s": round(st["executor_s"], 4), "MHz_with_trace_kept": round(cycles / st["executor_s"] / 1e6, 2), "host_ms_per_segment": round(1e3 * st["executor_s"] / max(1, st["segments"]), 3),
                         "note": "own host thread, overlaps the device work of the previous segment"},
            "device": {"witgen_ms_per_segment": round(st["witgen_ms"] / max(1, st["segments"]), 3), "prove_ms_per_segment": round(st["prove_ms"] / max(1, st["segments"]), 3),
                       "note": "witgen = upload of 72 B/cycle + 16 B/boundary row and the expansion kernel; prove = r0h_prove_segment_committed (CODE committed once per trace size)"},
            "circuit": "trace.r0c W=(%d accum, %d code, %d data)" % tuple(gc.group_size), "seal_words": int(seals[0][1].size),
            "receipt_verified": True, "journal_commitment": commitment(r0, receipt.journal), "data": "synthetic guest; no guest ELF exists in the reference (needs the Rust toolchain)"}
    print(json.dumps(line))
    gc.free()
    hal.close()
    if env is not None:
        env.close()


if __name__ == "__main__":
    main()
