#!/usr/bin/env python3
"""A/B timing of r0h_hash_rows / r0h_hash_fold across library variants (tools/ab/build_variants.sh) in one process on one box:
the variants take turns, several rounds, so that clock drift and box-to-box differences cancel.  HIP events from the library's
own kernel timing.  usage: ab_hash.py [--rows-po2 20] [--cols 192] [--rounds 3] name1 name2 ..."""
import argparse
import ctypes
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
_vp, _u32, _sz = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_size_t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows-po2", type=int, default=22)
    ap.add_argument("--cols", type=int, default=192)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--launches", type=int, default=6)
    ap.add_argument("names", nargs="+")
    a = ap.parse_args()
    rows = 1 << a.rows_po2
    libs = {}
    for n in a.names:
        L = ctypes.CDLL(os.path.join(HERE, "lib", "libr0hip_%s.so" % n))
        for f in ("r0h_ctx_create", "r0h_buf_alloc", "r0h_hash_rows", "r0h_hash_fold", "r0h_kernel_timing", "r0h_kernel_stats", "r0h_sync", "r0h_buf_h2d"):
            getattr(L, f).restype = _vp
        ctx = _vp()
        assert not L.r0h_ctx_create(0, ctypes.byref(ctx))
        mat, dig = _vp(), _vp()
        assert not L.r0h_buf_alloc(ctx, _sz(rows * a.cols * 4), ctypes.byref(mat))
        assert not L.r0h_buf_alloc(ctx, _sz(rows * 2 * 32), ctypes.byref(dig))
        # random canonical words (zeros would let the chip clock higher): one column pattern uploaded per 64 MiB chunk
        import numpy as np
        rng = np.random.default_rng(1)
        chunk = rng.integers(0, 2013265921, 1 << 24, dtype=np.uint32)
        for off in range(0, rows * a.cols * 4, chunk.nbytes):
            n_b = min(chunk.nbytes, rows * a.cols * 4 - off)
            assert not L.r0h_buf_h2d(ctx, mat, _sz(off), chunk.ctypes.data_as(_vp), _sz(n_b))
        libs[n] = (L, ctx, mat, dig)
    res = {n: {"hash_rows_ms": [], "hash_fold_ms": []} for n in a.names}
    for rnd in range(a.rounds + 1):
        for n in a.names:
            L, ctx, mat, dig = libs[n]
            assert not L.r0h_kernel_timing(ctx, 1)
            for _ in range(a.launches):
                err = L.r0h_hash_rows(ctx, dig, mat, _u32(rows), _u32(a.cols))
                assert not err, ctypes.cast(err, ctypes.c_char_p).value
                err = L.r0h_hash_fold(ctx, dig, _u32(rows // 2))
                assert not err
            buf = ctypes.create_string_buffer(1 << 14)
            assert not L.r0h_kernel_stats(ctx, buf, _sz(len(buf)))
            st = json.loads(buf.value.decode())
            if rnd:  # round 0 warms up
                res[n]["hash_rows_ms"].append(st["hash_rows_kernel"]["total_ms"] / a.launches)
                res[n]["hash_fold_ms"].append(st["hash_fold_kernel"]["total_ms"] / a.launches)
    for n in a.names:
        r = res[n]
        print("%-16s hash_rows %s  mean %.4f ms   hash_fold(2M parents) %s  mean %.4f ms" % (
            n, " ".join("%.4f" % x for x in r["hash_rows_ms"]), sum(r["hash_rows_ms"]) / len(r["hash_rows_ms"]),
            " ".join("%.4f" % x for x in r["hash_fold_ms"]), sum(r["hash_fold_ms"]) / len(r["hash_fold_ms"])))


if __name__ == "__main__":
    sys.exit(main())
