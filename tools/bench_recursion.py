#!/usr/bin/env python3
"""BASELINE.json configs[4] in shape: every rank proves its share of segments, lifts each seal, folds its own nodes, and the
ranks join pairwise up a binary tree over torch.distributed point-to-point (RCCL on a GPU node).  Rank 0 prints one JSON
line with the tree latency.  The recursion circuit is this repository's (hyperfridge-r0_amd/recursion.py): it computes the digest of what
a step consumes inside the proof (one Poseidon2 round per row); the consumed seals themselves are checked by the host-side verifier
beside the proof, not inside it.

  python tools/bench_recursion.py --segments 8                                    one GPU, whole tree in-process
  python tools/bench_recursion.py --gpus N --segments 64 [--backend gloo --share-device]
                                   N ranks, segments sharded: this process starts one fresh process per rank before touching the GPU
                                   (an outer `python -m torch.distributed.run ...` that sets WORLD_SIZE works too)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=0, help="ranks to start (one per GPU); 0 = take WORLD_SIZE from the environment, else 1")
    ap.add_argument("--segments", type=int, default=8, help="segment seals over all ranks")
    ap.add_argument("--circuit", default="bench", help="segment circuit (bench at po2 20 is configs[4]; small for tests)")
    ap.add_argument("--segment-po2", type=int, default=20)
    ap.add_argument("--recursion-po2", type=int, default=18)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--root-out", default="", help="rank 0 writes the root seal there (.npy) for checking")
    ap.add_argument("--share-device", action="store_true", help="all ranks use GPU 0 (rehearsal on a one-GPU box; needs --backend gloo)")
    args = ap.parse_args()

    import __graft_entry__ as entry
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import bench  # the same child-rank starter as bench.py: fresh processes, nothing re-exec'd after the GPU was touched
        entry.ensure_built()
        raise SystemExit(bench.spawn_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))

    import bench
    bench.die_with_parent()  # a rank started by spawn_ranks ends with its parent, whatever ends the parent
    bench.guard_stdout()  # stdout carries the one result line; what gloo / RCCL print to stdout themselves goes to stderr

    import numpy as np
    import torch

    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if local == 0:
        entry.ensure_built()
    if args.segments < world:
        raise SystemExit("bench_recursion.py: %d segments for %d ranks: every rank needs at least one" % (args.segments, world))
    device = 0 if args.share_device else local
    torch.cuda.set_device(device)
    torch.zeros(1, device="cuda")  # torch initialises HIP before libr0hip.so is loaded
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(args.backend, rank=rank, world_size=world,
                                device_id=torch.device("cuda", device) if args.backend == "nccl" else None)
    import hyperfridge_r0_amd as r0
    from hyperfridge_r0_amd import driver, recursion

    hal = r0.Hal(device)
    seg_blob = np.fromfile(entry.circuit_blob_path(args.circuit), dtype=np.uint32)
    rec_blob = np.fromfile(entry.circuit_blob_path("recursion"), dtype=np.uint32)
    seg_co = entry.code_object_path(args.circuit)
    seg = hal.load_circuit(seg_blob, seg_co if os.path.exists(seg_co) else None)
    # one session of args.segments segments: claims that chain (segment k ends in SystemSplit where k + 1 starts, the last one halts
    # with the journal's output); every segment seal is proved FOR its claim, every rank owns a contiguous run of them
    journal = r0.serde_encode_str('{"segments":%d}' % args.segments)
    claims, image_id = r0.session_claims(args.segments, journal)
    seg_cc = hal.code_commit(seg, args.segment_po2)
    rec = recursion.Recursor(hal, rec_blob, seg_blob, entry.code_object_path("recursion"), po2=args.recursion_po2, segment_roots={args.segment_po2: seg_cc.root()})
    mine = driver.shard_contiguous(args.segments, world, rank)

    def barrier():
        hal.sync()
        if world > 1:
            dist.barrier()

    # warm-up: one of everything (code objects loaded, pools filled)
    wclaims, _ = r0.session_claims(2, journal, state_seed=b"warm-up")
    warm = []
    for k in range(2):
        code, data, glob = hal.witgen(seg, args.segment_po2, 999 + k, globals_in=wclaims[k].globals())
        warm.append(hal.prove_segment(seg, args.segment_po2, seg_cc, data, glob))
        if k == 0:
            code.free(); data.free()
    rec.join(rec.lift(warm[0], wclaims[0]), rec.lift(warm[1], wclaims[1]))

    barrier()
    t0 = time.perf_counter()
    seals = []
    for s in mine:
        c2, d2, g2 = hal.witgen(seg, args.segment_po2, 1000 + s, globals_in=claims[s].globals())
        seals.append(hal.prove_segment(seg, args.segment_po2, seg_cc, d2, g2))
        c2.free(); d2.free()
    barrier()
    t1 = time.perf_counter()
    nodes = [rec.lift(seal, claims[s]) for seal, s in zip(seals, mine)]
    hal.sync()
    t_lifted = time.perf_counter()
    node = rec.fold(nodes) if nodes else None
    barrier()
    t2 = time.perf_counter()
    if world > 1:
        send, recv = recursion.torch_transport(torch.device("cuda", device))
        node = recursion.join_across_ranks(rec, node, rank, world, send, recv)
    barrier()
    t3 = time.perf_counter()
    if rank == 0:
        # the root: its seal verifies bound to the recursion circuit's control root, names its claim, and that claim is the session's
        # end-to-end claim (first pre-state = the image id, last post-state, Halted(0), the journal's output digest)
        end_to_end = r0.ReceiptClaim.make(claims[0].pre, claims[-1].post, claims[-1].exit_system, claims[-1].exit_user, bytes(claims[-1].output_digest))
        root_ok = rec.verify(node) and node.claim.digest() == end_to_end.digest() and node.claim.pre.digest() == image_id
        if args.root_out:
            np.save(args.root_out, node.seal)
            np.save(args.root_out + ".control_root.npy", rec.control_root)
            open(args.root_out + ".claim.bin", "wb").write(bytes(node.claim))
        steps = len(recursion.tree_schedule(world))
        bench.emit_result({
            "metric": "lift+join tree over segment seals (configs[4] in shape; this repository's recursion circuit with the in-circuit digest, see hyperfridge-r0_amd/recursion.py)",
            "n_gpus": world, "segments": args.segments, "segment_po2": args.segment_po2, "recursion_po2": args.recursion_po2,
            "prove_segments_s": round(t1 - t0, 4), "lift_and_local_fold_s": round(t2 - t1, 4),
            "ms_per_lift": round(1e3 * (t_lifted - t1) / max(1, len(mine)), 2), "ms_per_local_join": round(1e3 * (t2 - t_lifted) / max(1, len(mine) - 1), 2),
            "segment_seal_words": int(seals[0].size) if seals else 0, "cross_rank_joins_s": round(t3 - t2, 4),
            "cross_rank_join_steps": steps, "tree_latency_s": round(t3 - t1, 4), "end_to_end_s": round(t3 - t0, 4),
            "root_verifies": bool(root_ok), "root_claim_is_the_sessions_end_to_end_claim": bool(node.claim.digest() == end_to_end.digest()),
            "root_seal_words": int(node.seal.size), "backend": args.backend if world > 1 else "none",
            "data": "synthetic"})
    code.free(); data.free(); seg_cc.free()
    rec.close(); seg.free()
    hal.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
