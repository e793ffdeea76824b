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


LARGE = [("small", 21), ("bench", 21), ("small", 22), ("bench", 22), ("small", 23), ("tiny", 24), ("bench", 23)]


def main():
    budget = float(sys.argv[1]) * 60 if len(sys.argv) > 1 else 120.0
    orc, hal = orc_binding.load(), r0.Hal(0)
    cases, t0, n = [("tiny", p) for p in (9, 10, 11, 12, 13)] + [("small", p) for p in (9, 10, 11, 12)] + [("recursion", 10), ("bench", 9), ("bench", 10)], time.time(), 0
    full = large = 0
    loaded = {}
    roots = {}  # (circuit, po2) -> control root: every seal is verified bound to its program, by both verifiers
    seed = 10_000

    def root_of(name, gc, po2):
        if (name, po2) not in roots:
            roots[(name, po2)] = hal.code_root(gc, po2)
        return roots[(name, po2)]

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
            root = root_of(name, gc, po2)
            assert r0.verify_seal(blob, seal, code_root=root)[0] == 0 and oc.verify(seal, code_root=root)[0] == 0
            code.free(); data.free()
            n += 1
        # one full-size segment per sweep: too large for the oracle's prover, so both verifiers must accept it
        blob, oc, gc = loaded["bench"]
        seed += 1
        code, data, glob = hal.witgen(gc, 20, seed)
        seal = hal.prove_segment(gc, 20, code, data, glob)
        code.free(); data.free()
        root = root_of("bench", gc, 20)
        if r0.verify_seal(blob, seal, code_root=root)[0] != 0 or oc.verify(seal, code_root=root)[0] != 0:
            print("FULL-SIZE seal rejected, seed %d" % seed)
            sys.exit(1)
        full += 1
        # and one segment above the default size (three-level transforms from 2^22 rows on), rotating
        name, po2 = LARGE[large % len(LARGE)]
        if name not in loaded:
            blob = np.fromfile(os.path.join(ROOT, "circuits", name + ".r0c"), dtype=np.uint32)
            loaded[name] = (blob, orc.circuit(blob), hal.load_circuit(blob))
        blob, oc, gc = loaded[name]
        seed += 1
        code, data, glob = hal.witgen(gc, po2, seed)
        seal = hal.prove_segment(gc, po2, code, data, glob)
        root = hal.code_root(gc, po2, code)
        code.free(); data.free()
        if r0.verify_seal(blob, seal, code_root=root)[0] != 0 or oc.verify(seal, code_root=root)[0] != 0:
            print("LARGE seal rejected, circuit %s po2 %d seed %d" % (name, po2, seed))
            sys.exit(1)
        large += 1
        print("%d seals identical, %d full-size and %d larger seals accepted by both verifiers after %.0f s" % (n, full, large, time.time() - t0), flush=True)
    print("soak ok: %d seals device == oracle word for word; %d seals at 2^20 rows and %d at 2^21..2^24 rows accepted by both verifiers; all bound to their control roots"
          % (n, full, large))


if __name__ == "__main__":
    main()
