#!/usr/bin/env python3
"""Experiment driver for the eval_check code generator.
  emit:  python tools/tune_evalcheck.py emit NAME SCOPE WAVES BUDGET   -> build/ec/NAME.hip (compile with hipcc --genco)
  time:  python tools/tune_evalcheck.py time PO2 NAME [NAME...]        -> times r0h_eval_check with each code object (GPU)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import hyperfridge_r0_amd as r0

OUT = os.path.join(ROOT, "build", "ec")


def emit(name, scope, waves, budget, circuit="bench"):
    os.environ["R0H_EC_SCOPE"], os.environ["R0H_EC_WAVES"], os.environ["R0H_EC_BUDGET"] = scope, waves, budget
    if name.startswith("fuse"):  # sums of products share one reduction (circuit.hip plan_fusion); off by default
        os.environ["R0H_EC_FUSION"] = "1"
    else:
        os.environ.pop("R0H_EC_FUSION", None)
    blob = np.fromfile(os.path.join(ROOT, "circuits", circuit + ".r0c"), dtype=np.uint32)
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, name + ".hip"), "w") as f:
        f.write(r0.emit_eval_check_source(blob))


def timeit(po2, names, circuit="bench"):
    hal = r0.Hal(0)
    blob = np.fromfile(os.path.join(ROOT, "circuits", circuit + ".r0c"), dtype=np.uint32)
    base = hal.load_circuit(blob, os.path.join(OUT, names[0] + ".hsaco"))
    code, data, glob = hal.witgen(base, po2, 1)
    n, dom = 1 << po2, 4 << po2
    rng = np.random.default_rng(1)
    mix = rng.integers(0, r0.P, base.n_mix, dtype=np.uint32)
    pm = rng.integers(0, r0.P, 4, dtype=np.uint32)
    accum = hal.accum(base, po2, code, data, mix)
    ev = []
    for buf, cnt in ((accum, base.group_size[0]), (code, base.group_size[1]), (data, base.group_size[2])):
        e = hal.alloc(cnt * dom)
        hal.batch_expand_into_evaluate_ntt(e, buf, cnt, po2, 2)  # any well-mixed field data will do for timing
        ev.append(e)
    ref = None
    for name in names:
        c = hal.load_circuit(blob, os.path.join(OUT, name + ".hsaco"))
        chk = hal.eval_check(c, po2, ev[0], ev[1], ev[2], glob, mix, pm)
        hal.sync()
        out = chk.to_host()
        if ref is None:
            ref = out
        same = bool(np.array_equal(out, ref))
        hal.kernel_timing(True)
        t0 = time.perf_counter()
        for _ in range(3):
            chk2 = hal.eval_check(c, po2, ev[0], ev[1], ev[2], glob, mix, pm)
            chk2.free()
        hal.sync()
        wall = (time.perf_counter() - t0) / 3
        st = hal.kernel_stats()["eval_check"]
        hal.kernel_timing(False)
        print("%-28s eval_check %.2f ms (wall %.2f ms)  identical_to_first=%s" % (name, st["total_ms"] / st["launches"], wall * 1e3, same), flush=True)
        chk.free()
        c.free()
    hal.close()


if __name__ == "__main__":
    if sys.argv[1] == "emit":
        emit(*sys.argv[2:6])
    else:
        timeit(int(sys.argv[2]), sys.argv[3:])
