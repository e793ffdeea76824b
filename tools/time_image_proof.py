#!/usr/bin/env python3
"""Time of one image proof (r0h_prove_image) for the camt53 guest on the GPU box, with the prover's phases, and of the host-side witness."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import __graft_entry__ as e
import hyperfridge_r0_amd as r0
import guest_camt53
elf, stream, what = guest_camt53.elf_and_input()
hal = r0.Hal(0)
ic = hal.load_circuit(np.fromfile(e.circuit_blob_path("image"), dtype=np.uint32), e.code_object_path("image"))
ch = np.arange(16, dtype=np.uint32) + 5
for i in range(6):
    t0 = time.perf_counter(); s = hal.prove_image(ic, elf, ch); dt = time.perf_counter() - t0
    print("prove_image %.2f ms, %d words, po2 %d" % (1e3 * dt, s.size, r0.image_po2(elf)), flush=True)
    print("  phases:", ", ".join("%s=%.2f" % p for p in hal.last_profile()))
t0 = time.perf_counter(); r0.image_witness(elf); print("host witness %.2f ms" % (1e3 * (time.perf_counter() - t0)))
