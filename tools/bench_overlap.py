#!/usr/bin/env python3
"""Does an HBM-bound kernel (NTT) overlap with a VALU-bound one (hash_rows) when issued from two contexts/streams?"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import hyperfridge_r0_amd as r0

def main():
    po2, n, dom, cols = 20, 1 << 20, 1 << 22, 192
    rng = np.random.default_rng(0)
    hs = [r0.Hal(0), r0.Hal(0)]
    st = []
    for h in hs:
        src = h.copy_from(rng.integers(0, r0.P, cols * n, dtype=np.uint32))
        ev = h.alloc(cols * dom)
        h.batch_expand_into_evaluate_ntt(ev, src, cols, po2, 2)
        dig = h.alloc(dom * 8)
        h.sync()
        st.append((src, ev, dig))
    def hash_job(i, reps):
        h, (src, ev, dig) = hs[i], st[i]
        for _ in range(reps):
            h.hash_rows(dig, ev, dom, cols)
        h.sync()
    def ntt_job(i, reps):
        h, (src, ev, dig) = hs[i], st[i]
        for _ in range(reps):
            h.batch_expand_into_evaluate_ntt(ev, src, cols, po2, 2)
        h.sync()
    def wall(jobs):
        ts = [threading.Thread(target=f, args=a) for f, a in jobs]
        t0 = time.perf_counter()
        for t in ts: t.start()
        for t in ts: t.join()
        return (time.perf_counter() - t0) * 1e3
    wall([(hash_job, (0, 1)), (ntt_job, (1, 1))])
    a = wall([(hash_job, (0, 4))]); b = wall([(ntt_job, (1, 16))])
    both = wall([(hash_job, (0, 4)), (ntt_job, (1, 16))])
    hh = wall([(hash_job, (0, 4)), (hash_job, (1, 4))])
    print("hash x4 alone %.1f ms | ntt x16 alone %.1f ms | together %.1f ms (sum %.1f) | hash||hash %.1f ms" % (a, b, both, a + b, hh))
    for h in hs: h.close()
main()
