#!/usr/bin/env python3
"""Soak run on the GPU box: many seeds x circuits x sizes, device seal against the oracle's, word for word.
usage: python tools/soak_parity.py [minutes]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import hyperfridge_r0_amd as r0
import orc_binding


def main():
    budget = float(sys.argv[1]) * 60 if len(sys.argv) > 1 else 120.0
    orc, hal = orc_binding.load(), r0.Hal(0)
    cases, t0, n = [("tiny", p) for p in (9, 10, 11, 12)] + [("small", p) for p in (9, 10, 11)] + [("recursion", 10)], time.time(), 0
    loaded = {}
    seed = 10_000
    while time.time() - t0 < budget:
        for name, po2 in cases:
            if name not in loaded:
                blob = np.fromfile(os.path.join(ROOT, "circuits", name + ".r0c"), dtype=np.uint32)
                loaded[name] = (blob, orc.circuit(blob), hal.load_circuit(blob))
            blob, oc, gc = loaded[name]
            seed += 1
            code, data, glob = hal.witgen(gc, po2, seed)
            seal = hal.prove_segment(gc, po2, code, data, glob)
            ocode, odata, oglob = oc.witgen(po2, seed)
            want = oc.prove(po2, ocode, odata, oglob)
            if not np.array_equal(seal, want):
                bad = int(np.nonzero(seal[:min(seal.size, want.size)] != want[:min(seal.size, want.size)])[0][0]) if seal.size and want.size else -1
                print("MISMATCH circuit %s po2 %d seed %d at word %d" % (name, po2, seed, bad))
                sys.exit(1)
            assert r0.verify_seal(blob, seal)[0] == 0
            code.free(); data.free()
            n += 1
        print("%d seals identical after %.0f s" % (n, time.time() - t0), flush=True)
    print("soak ok: %d seals, device == oracle word for word" % n)


if __name__ == "__main__":
    main()
