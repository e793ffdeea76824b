#!/usr/bin/env python3
"""Write tests/golden/seal_<circuit>_po2_<n>_seed_<s>.npy with the CPU oracle: frozen outputs of the composed protocol, so
that a change to the transcript, the seal layout or any operation shows up as a golden mismatch (in the oracle's CPU test and
in the GPU parity test) instead of passing silently because oracle and product changed together."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc_binding

CASES = [("tiny", 9, 1)]


def main():
    orc = orc_binding.load()
    for name, po2, seed in CASES:
        blob = np.fromfile(os.path.join(ROOT, "circuits", name + ".r0c"), dtype=np.uint32)
        c = orc.circuit(blob)
        code, data, glob = c.witgen(po2, seed)
        seal = c.prove(po2, code, data, glob)
        assert c.verify(seal) == (0, "ok")
        path = os.path.join(ROOT, "tests", "golden", "seal_%s_po2_%d_seed_%d.npy" % (name, po2, seed))
        np.save(path, seal)
        print(path, seal.size, "words")


if __name__ == "__main__":
    main()
