"""Multi-GPU driver logic for segment-parallel proving (one process per GPU, no data-path collective).

Segments are independent units: rank r proves segments r, r + world, r + 2*world, ...  The only
communication is the barrier around the timed region and the MAX / SUM reductions of the per-rank results, over
torch.distributed (backend "nccl" = RCCL on the GPU box, "gloo" in the CPU tests).

Order matters in a process that uses both: import torch (and set the device, build the process group) BEFORE libr0hip.so is first
used -- the library then binds to the HIP runtime torch has loaded; the other way round the process holds two HIP runtimes and the
second one to initialise finds no device (seen on the GPU box under torch.distributed.run; bench.py and tools/bench_session.py keep
this order)."""
import os
import time


def shard_segments(total, world, rank):
    """Static round-robin partition of `total` segment indices (BASELINE.json config 3)."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world of %d" % (rank, world))
    return list(range(rank, total, world))


def shard_contiguous(total, world, rank):
    """Balanced partition into contiguous runs, rank order = segment order: what the lift/join tree needs (a join composes two
    claims that follow one another, so a rank's nodes and its neighbour's must be neighbours in the session)."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world of %d" % (rank, world))
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return list(range(start, start + base + (1 if rank < extra else 0)))


class DistEnv:
    """Rank bookkeeping + the three collectives the driver needs.  backend=None means single process."""

    def __init__(self, backend=None, device=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dist = None
        self.device = device
        # R0H_FORCE_PROCESS_GROUP=1: build the process group even for one rank, so that the backend's path (RCCL init with a device id,
        # barrier, device-tensor all-reduce) can be exercised on a one-GPU box (tests/test_gpu_bench.py)
        if self.world > 1 or (backend is not None and os.environ.get("R0H_FORCE_PROCESS_GROUP") == "1"):
            if backend is None:
                raise ValueError("WORLD_SIZE > 1 needs a backend")
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            if not dist.is_initialized():
                kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
                dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world, **kw)
            self.dist = dist

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def _reduce(self, value, op_name):
        if self.dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64, device=self.device if self.device is not None else "cpu")
        self.dist.all_reduce(t, op=getattr(self.dist.ReduceOp, op_name))
        return float(t.item())

    def max(self, value):
        return self._reduce(value, "MAX")

    def sum(self, value):
        return self._reduce(value, "SUM")

    def sum_array(self, array):
        """element-wise sum over the ranks of an int64 numpy array (all-reduce; the array itself with one rank)"""
        if self.dist is None:
            return array
        import numpy as np
        import torch
        t = torch.from_numpy(np.ascontiguousarray(array, dtype=np.int64))
        if self.device is not None:
            t = t.to(self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t.cpu().numpy()

    def close(self):
        if self.dist is not None and self.dist.is_initialized():
            self.dist.destroy_process_group()


def run_timed(env, step_fn, steps, warmup, device_sync=lambda: None, many_fn=None):
    """`warmup` untimed calls of step_fn(i), then exactly `steps` timed ones bracketed by barrier + device sync on both
    sides; returns (max-over-ranks seconds, units processed by all ranks), where step_fn returns the units it processed.
    many_fn(n), when given, performs n steps in one call (e.g. every in-flight lane running its n segments back to back,
    without a join between steps) and returns the units; it replaces the per-step loop for both phases."""
    if many_fn is not None:
        if warmup:
            many_fn(warmup)
    else:
        for i in range(warmup):
            step_fn(-1 - i)
    device_sync()
    env.barrier()
    t0 = time.perf_counter()
    units = 0
    if many_fn is not None:
        units = int(many_fn(steps) or 0)
    for i in range(steps if many_fn is None else 0):
        units += int(step_fn(i) or 0)
    device_sync()
    env.barrier()
    elapsed = time.perf_counter() - t0
    return env.max(elapsed), env.sum(units)


def prove_elf_sharded(env, hal, circuit, elf, input_words, segment_po2=20, max_cycles=0):
    """`prove(env, elf)` on env.world GPUs: every rank executes the guest (deterministic, a tenth of a second per ten million cycles)
    and commits segments rank, rank + world, ... (r0h_session_begin); with the trace circuit the segments of a session share ONE
    challenge derived from all their DATA roots, so the ranks exchange their records -- 28 words per segment, one all-reduce of a
    table every rank fills its own rows of (RCCL under "nccl", gloo otherwise: the one collective of the path) -- and finish their
    proofs under it (r0h_session_finish).  The ranks' receipts travel to rank 0 as JSON over point-to-point send / recv
    (hyperfridge-r0_amd/recursion.py torch_transport) and are merged there (r0h_receipt_merge): the merged receipt is the one a single
    GPU would have produced, seal for seal.  Returns (Receipt on rank 0 / None elsewhere, image id, cycles)."""
    import numpy as np
    import hyperfridge_r0_amd as r0
    ses = hal.session_begin(circuit, elf, input_words, segment_po2=segment_po2, max_cycles=max_cycles, part=env.rank, parts=env.world)
    idx, rec = ses.records()
    table = np.zeros((ses.n_segments, r0.SESSION_RECORD_WORDS), dtype=np.int64)
    if len(idx):
        table[idx] = rec
    table = env.sum_array(table)  # every row is written by exactly one rank (words below 2^31: the sum is exact)
    mine, image_id, cycles = ses.finish(table.astype(np.uint32))
    ses.close()
    if env.world == 1:
        return mine, image_id, cycles
    from .recursion import torch_transport
    send, recv = torch_transport(env.device)
    if env.rank != 0:
        text = mine.to_json().encode()
        send(np.frombuffer(text + bytes(-len(text) % 4), dtype=np.uint32), 0)
        send(np.array([len(text)], dtype=np.uint32), 0)
        return None, image_id, cycles
    parts = [mine]
    for src in range(1, env.world):
        words = recv(src)
        n = int(recv(src)[0])
        parts.append(r0.Receipt.parse(words.tobytes()[:n].decode()))
    return r0.Receipt.merge(parts), image_id, cycles
