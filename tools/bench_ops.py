#!/usr/bin/env python3
"""Per-operation timing through the C ABI at bench sizes (po2 = 20): used to A/B kernel variants on the GPU box."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import hyperfridge_r0_amd as r0

if os.environ.get("R0HIP_AB_LIB"):  # A/B runs: a variant build of the library (e.g. build/varB/libr0hip.so)
    r0.LIB_PATH = os.path.abspath(os.environ["R0HIP_AB_LIB"])


def main():
    ops = sys.argv[1:] or ["hash_rows", "hash_fold", "intt", "ntt", "bitrev", "zk", "evalany"]
    hal = r0.Hal(0)
    po2, n, dom = 20, 1 << 20, 1 << 22
    rng = np.random.default_rng(0)
    cols = 192
    src = hal.copy_from(rng.integers(0, r0.P, cols * n, dtype=np.uint32))
    ev = hal.alloc(cols * dom)
    hal.batch_expand_into_evaluate_ntt(ev, src, cols, po2, 2)
    nodes = hal.alloc(dom * 2 * 8)
    hal.sync()

    def timed(name, fn, reps=3):
        fn()
        hal.sync()
        hal.kernel_timing(True)
        for _ in range(reps):
            fn()
        st = hal.kernel_stats()
        hal.kernel_timing(False)
        print("%-10s " % name + "  ".join("%s=%.3f ms" % (k, v["total_ms"] / reps) for k, v in st.items() if v["launches"]), flush=True)

    if "hash_rows" in ops:
        timed("hash_rows", lambda: hal.hash_rows(nodes.slice(dom * 8, dom * 8), ev, dom, cols))
    if "hash_fold" in ops:
        def folds():
            sz = dom // 2
            while sz >= 1:
                hal.hash_fold(nodes, sz)
                sz //= 2
        timed("hash_fold", folds)
    if "intt" in ops:
        timed("intt", lambda: hal.batch_interpolate_ntt(src, cols, po2))
    if "ntt" in ops:
        timed("ntt", lambda: hal.batch_expand_into_evaluate_ntt(ev, src, cols, po2, 2))
    if "bitrev" in ops:
        timed("bitrev", lambda: hal.batch_bit_reverse(src, cols, po2))
    if "zk" in ops:
        timed("zk", lambda: hal.zk_shift(src, cols, po2))
    if "evalany" in ops:
        which = np.arange(cols, dtype=np.uint32)
        xs = np.tile(rng.integers(0, r0.P, 4, dtype=np.uint32), cols)
        out = hal.alloc(4 * cols)
        timed("evalany", lambda: hal.batch_evaluate_any(src, po2, which, xs, out))
    hal.close()


if __name__ == "__main__":
    main()
