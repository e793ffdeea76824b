"""CPU tests of the N > 1 path: the bench driver's sharding, barrier and reductions with gloo, world_size 2.
The proving step itself needs a GPU, so a stand-in step (sleep + unit count) is injected; what is under test is the
multi-process logic bench.py runs around r0h_prove_segment."""
import os
import socket
import sys
import time

import pytest
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, total_segments, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import hyperfridge_r0_amd  # noqa: F401  (package import must work without a GPU)
    from hyperfridge_r0_amd import driver
    env = driver.DistEnv(backend="gloo")
    mine = driver.shard_segments(total_segments, env.world, env.rank)
    proved = []

    def step(i):
        if i < 0:
            return 0
        time.sleep(0.02 * (1 + env.rank))  # rank 1 is slower: the MAX must pick it up
        batch = mine[i::3]
        proved.extend(batch)
        return len(batch)

    elapsed, units = driver.run_timed(env, step, steps=3, warmup=1)
    out.put((rank, elapsed, units, sorted(proved)))
    env.close()


def _session(n=7):
    """seven segments of a made-up session: (journal, seals, claims)"""
    import numpy as np
    import hyperfridge_r0_amd as r0
    states = [r0.SystemState.make(0x1000 + 4 * i, bytes([i + 1]) * 32) for i in range(n + 1)]
    claims = [r0.ReceiptClaim.make(states[i], states[i + 1], 2 if i < n - 1 else 0, 0, None) for i in range(n)]
    return r0.serde_encode_str('{"iban":"X"}'), [np.arange(300, dtype=np.uint32) * (i + 3) for i in range(n)], claims


def _record(i):
    """the record segment i of the made-up session contributes to the session challenge (28 field words)"""
    import numpy as np
    return (np.arange(28, dtype=np.int64) * 7919 + 104729 * (i + 1)) % 2013265921


class _ShareOnlySession:
    def __init__(self, part, parts):
        self.part, self.parts, self.n_segments = part, parts, len(_session()[1])

    def records(self):
        import numpy as np
        idx = np.arange(self.part, self.n_segments, self.parts)
        return idx, np.stack([_record(i) for i in idx]).astype(np.uint32)

    def finish(self, all_records):
        import numpy as np
        import hyperfridge_r0_amd as r0
        # the exchange brought every rank's rows: the whole table, in index order, on every rank
        assert np.array_equal(np.asarray(all_records, dtype=np.int64), np.stack([_record(i) for i in range(self.n_segments)]))
        journal, seals, claims = _session()
        return r0.Receipt.new(journal, seals[self.part::self.parts], claims[self.part::self.parts], indices=list(range(self.part, len(seals), self.parts))), bytes(32), 123

    def close(self):
        pass


class _ShareOnlyHal:
    """stands where Hal.session_begin stands (the proofs need a GPU): hands out this rank's share of the made-up session"""
    def session_begin(self, circuit, elf, input_words, segment_po2=20, max_cycles=0, part=0, parts=1):
        return _ShareOnlySession(part, parts)


def _sharded_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from hyperfridge_r0_amd import driver
    env = driver.DistEnv(backend="gloo")
    receipt, _, cycles = driver.prove_elf_sharded(env, _ShareOnlyHal(), None, b"elf", [1, 2, 3])
    out.put((rank, None if receipt is None else receipt.to_json(), cycles))
    env.close()


def test_a_session_sharded_over_two_ranks_arrives_whole_on_rank_zero():
    """driver.prove_elf_sharded over gloo, world_size 2: rank r holds segments r, r + 2, ...; between the two phases of the session
    the ranks exchange their segments' records (one all-reduce of a table every rank fills its own rows of: each rank then holds
    the whole table); rank 1's receipt travels to rank 0 as JSON over send / recv and the merged receipt is the whole session's
    (the proofs themselves are stood in for: they need a GPU)."""
    import hyperfridge_r0_amd as r0
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(out.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    journal, seals, claims = _session()
    assert results[0][1] == r0.Receipt.new(journal, seals, claims).to_json() and results[1][1] is None and results[0][2] == results[1][2] == 123


def test_shard_segments_partitions_exactly():
    sys.path.insert(0, ROOT)
    from hyperfridge_r0_amd import driver
    for total in (0, 1, 7, 64):
        for world in (1, 2, 8):
            parts = [driver.shard_segments(total, world, r) for r in range(world)]
            assert sorted(x for p in parts for x in p) == list(range(total))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    with pytest.raises(ValueError):
        driver.shard_segments(4, 2, 2)


def test_two_rank_gloo_driver_aggregates_units_and_takes_the_slowest_rank():
    world, total = 2, 11
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, out)) for r in range(world)]
    for p in procs:
        p.start()
    results = [out.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    results.sort()
    (r0, e0, u0, p0), (r1, e1, u1, p1) = results
    assert e0 == e1 and u0 == u1 == total              # every rank sees the same reduced numbers
    assert sorted(p0 + p1) == list(range(total))        # all segments proved exactly once
    assert e0 >= 3 * 0.04 - 1e-3                        # the slow rank's time (3 steps x 40 ms) bounds the result


def test_run_timed_with_a_many_steps_callable():
    """many_fn(n) stands for n steps in one call (lanes running back to back): warm-up and the timed region each call it once."""
    from hyperfridge_r0_amd import driver
    calls = []
    env = driver.DistEnv(backend=None)
    elapsed, units = driver.run_timed(env, lambda i: 1 / 0, 5, 2, many_fn=lambda n: calls.append(n) or 3 * n)
    assert calls == [2, 5] and units == 15 and elapsed >= 0
    calls.clear()
    driver.run_timed(env, lambda i: 1 / 0, 4, 0, many_fn=lambda n: calls.append(n) or n)
    assert calls == [4]


def _bench(*argv, timeout=240):
    import json
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), capture_output=True, text=True, cwd=ROOT, timeout=timeout,
                         env={k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")})
    lines = [json.loads(ln) for ln in out.stdout.splitlines() if ln.startswith("{")]
    return out, lines


def test_bench_py_gpus_n_starts_its_own_ranks():
    """`python bench.py --gpus 2` as typed (no torchrun around it): the parent starts two fresh ranks before touching torch or the
    GPU, they rendezvous on 127.0.0.1, run the barrier-bracketed steps, and rank 0's single JSON line comes back through the
    parent.  The proving step needs a GPU, so the harness is rehearsed with a sleeping step (labelled as such)."""
    out, lines = _bench("--gpus", "2", "--steps", "3", "--warmup", "1", "--contexts", "2", "--rehearse-without-gpu", "30")
    assert out.returncode == 0, out.stderr[-2000:]
    assert len(lines) == 1
    # stdout carries that line and nothing else: what gloo (and RCCL) print to stdout themselves went to stderr (bench.guard_stdout)
    assert out.stdout.strip().splitlines() == [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    d = lines[0]
    assert d["n_gpus"] == 2 and d["steps"] == 3 and "REHEARSAL" in d["metric"]
    assert 25.0 < d["ms_per_step"] < 200.0
    assert abs(d["value"] - 2 * 2 * 3 / (d["ms_per_step"] * 3 / 1e3)) / d["value"] < 0.02  # units of all ranks / MAX time


def test_bench_py_reports_a_failed_rank_with_a_non_zero_exit():
    out, lines = _bench("--gpus", "3", "--steps", "2", "--warmup", "0", "--rehearse-without-gpu", "10", "--fail-rank", "2")
    assert out.returncode != 0 and lines == [] and "rank(s) failed" in out.stderr


@pytest.mark.parametrize("how", ["SIGTERM", "SIGKILL"])
def test_no_rank_outlives_a_parent_that_is_killed(how):
    """atexit alone does not run when the parent of the self-started ranks gets SIGTERM (a harness time limit) or SIGKILL;
    the ranks would sit in a barrier holding their GPUs.  The parent now forwards SIGTERM / SIGINT / SIGHUP to its ranks, and every
    rank arms PR_SET_PDEATHSIG before touching torch, which covers SIGKILL.  Rehearsed without a GPU: three ranks with long steps."""
    import signal
    import subprocess
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    parent = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "200", "--warmup", "0", "--rehearse-without-gpu", "500"],
                              cwd=ROOT, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)

    def ranks():
        out = subprocess.run(["ps", "-eo", "pid,ppid,args"], capture_output=True, text=True).stdout.splitlines()
        return [int(ln.split()[0]) for ln in out if len(ln.split()) > 2 and ln.split()[1] == str(parent.pid) and "bench.py" in ln]

    deadline = time.time() + 60
    kids = []
    while time.time() < deadline and len(kids) < 3:
        time.sleep(0.2)
        kids = ranks()
    assert len(kids) == 3, kids
    time.sleep(1.0)  # let them get into their steps (and past prctl)
    parent.send_signal(getattr(signal, how))
    parent.wait(timeout=30)
    deadline = time.time() + 20

    def alive(pid):
        try:
            os.kill(pid, 0)
        except OSError:
            return False
        try:  # a zombie waiting for init to reap it is dead for our purposes
            return open("/proc/%d/stat" % pid).read().split(")")[1].split()[0] != "Z"
        except OSError:
            return False

    while time.time() < deadline and any(alive(k) for k in kids):
        time.sleep(0.2)
    left = [k for k in kids if alive(k)]
    for k in left:
        os.kill(k, signal.SIGKILL)  # exactly the processes this test started
    assert left == [], "ranks that outlived their parent: %s" % left
