#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc counter_collection CSVs (SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES, and
SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES) into profiles/<round>/pmc_valu.json: per kernel family, the
vector-instruction count per wave and the share of wave time spent waiting for an issue slot vs waiting on memory.
usage: summarize_pmc_valu.py <pass1.csv> <pass2.csv> <out.json> "<how it was collected>" """
import collections
import csv
import json
import re
import sys


def load(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("r0h::", "")
        name = re.sub(r"^eval_check_\d+$", "eval_check_k", name)
        agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
    return agg


def main():
    a, b = load(sys.argv[1]), load(sys.argv[2])
    out = {"_about": sys.argv[4] + "  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves (MI355X_MICROARCH.md, "
                     "PMC table); WAIT_INST_ANY = a wave has an instruction ready but no issue slot (VALU pipe taken by another wave of the "
                     "SIMD), WAIT_ANY = parked on s_waitcnt / barrier (memory, LDS).  WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES.",
           "kernels": {}}
    for k in sorted(a, key=lambda k: -a[k]["SQ_WAVE_CYCLES"]):
        if k not in b or not b[k]["SQ_WAVES"]:
            continue
        wc, waves = a[k]["SQ_WAVE_CYCLES"], b[k]["SQ_WAVES"]
        tot = b[k]["SQ_WAIT_ANY"] + b[k]["SQ_WAIT_INST_ANY"] + b[k]["SQ_ACTIVE_INST_ANY"]
        out["kernels"][k] = {
            "waves": int(waves),
            "valu_insts_per_wave": round(a[k]["SQ_INSTS_VALU"] / waves, 1),
            "wave_quad_cycles_per_wave": round(wc / waves, 1),
            "share_issue_wait": round(b[k]["SQ_WAIT_INST_ANY"] / tot, 3),
            "share_memory_or_barrier_wait": round(b[k]["SQ_WAIT_ANY"] / tot, 3),
            "share_issuing": round(b[k]["SQ_ACTIVE_INST_ANY"] / tot, 3),
            "busy_cycles_sum_over_shader_engines": int(a[k]["SQ_BUSY_CYCLES"]),
        }
    json.dump(out, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
