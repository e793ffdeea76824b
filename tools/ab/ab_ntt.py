#!/usr/bin/env python3
"""A/B of the NTT kernels between library variants (tools/ab/build_variants.sh with UNIT=ntt): per-kernel time of the inverse and the
expanding forward transform at the sizes a 2^20-row segment uses, and a digest of the outputs so that two variants can be compared
word for word (the default build is the one the parity tests cover).  usage: [R0HIP_AB_LIB=tools/ab/lib/libr0hip_X.so] ab_ntt.py [cols]"""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np

import hyperfridge_r0_amd as r0

if os.environ.get("R0HIP_AB_LIB"):
    r0.LIB_PATH = os.path.abspath(os.environ["R0HIP_AB_LIB"])


def main():
    cols = int(sys.argv[1]) if len(sys.argv) > 1 else 184
    hal = r0.Hal(0)
    rng = np.random.default_rng(0)
    digest = hashlib.sha256()
    for po2, width in ((13, 5), (16, 3), (18, 3), (20, 2)):  # every kernel shape: one pass, two passes, the wide tiles
        n = 1 << po2
        src = hal.copy_from(rng.integers(0, r0.P, width * n, dtype=np.uint32))
        ev = hal.alloc(width * 4 * n)
        hal.batch_interpolate_ntt(src, width, po2)
        hal.batch_expand_into_evaluate_ntt(ev, src, width, po2, 2)
        digest.update(src.to_host().tobytes())
        digest.update(ev.to_host().tobytes())
        src.free(); ev.free()
    po2, n = 20, 1 << 20
    src = hal.copy_from(rng.integers(0, r0.P, cols * n, dtype=np.uint32))
    ev = hal.alloc(cols * 4 * n)
    for name, fn in (("intt", lambda: hal.batch_interpolate_ntt(src, cols, po2)), ("ntt", lambda: hal.batch_expand_into_evaluate_ntt(ev, src, cols, po2, 2))):
        fn()
        hal.sync()
        hal.kernel_timing(True)
        for _ in range(5):
            fn()
        st = hal.kernel_stats()
        hal.kernel_timing(False)
        print("%-5s %d columns of 2^20: " % (name, cols) + "  ".join("%s=%.3f ms" % (k, v["total_ms"] / 5) for k, v in st.items() if v["launches"]), flush=True)
    print("lib", os.path.basename(r0.LIB_PATH), "outputs sha256", digest.hexdigest())
    hal.close()


if __name__ == "__main__":
    main()
