#!/usr/bin/env python3
"""Static VALU issue floor of one Poseidon2 permutation as compiled for gfx950.

Compiles tools/microbench/p2_rounds_bench.hip to assembly, walks the permutation kernel (first linear layer, the two loops
of four full rounds, the straight-line partial phase), counts the vector instructions one permutation executes and prices
them at the issue costs measured by tools/microbench/valu_rate_bench.hip (profiles/r01/valu_rate_microbench.txt):
  2.4 SIMD cycles per wave instruction   v_add/sub/xor/shift/mov/cndmask/sub_co (full rate: the best rate measured for any of them)
  4.3                                    v_mul_lo/hi_u32 and every other integer VOP3 (half rate)
  5.5                                    v_mad_u64_u32 with a zero addend
  8.1                                    v_mad_u64_u32 with a register addend
That sum is an ESTIMATE ("rate_table_estimate"): the rates were measured on isolated streams whose clock under load need not be
the kernel's, and round 1 showed the kernel beating such a sum by 2 %.  The FLOOR written beside it uses only what the hardware
guarantees: a wave64 instruction takes at least 2 cycles on a SIMD-32, and no 32-bit multiply (v_mul_lo/hi, v_mad_u64_u32 in either
form) issues faster than half rate, 4 cycles.  bench.py divides that floor by the cycles the measured time can hold at the 2.4 GHz
maximum clock, so roofline.valu_view.frac <= 1 by construction.
Writes profiles/<round>/p2_issue_floor.json; bench.py reads the newest for roofline.valu_view.
usage: p2_issue_floor.py [out.json]"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FULL = ("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_xor_b32", "v_and_b32", "v_or_b32", "v_lshlrev_b32", "v_lshrrev_b32", "v_ashrrev_i32",
        "v_mov_b32", "v_cndmask_b32", "v_sub_co_u32", "v_subrev_co_u32", "v_add_co_u32", "v_addc_co_u32", "v_subb_co_u32")
RATE = {"full_rate": 2.4, "half_rate": 4.3, "mad64_zero_addend": 5.5, "mad64_register_addend": 8.1}   # measured, isolated streams
FLOOR = {"full_rate": 2.0, "half_rate": 4.0, "mad64_zero_addend": 4.0, "mad64_register_addend": 4.0}  # guaranteed lower bounds


def classify(op, line):
    base = re.sub(r"_e(32|64)$", "", op)
    if base == "v_mad_u64_u32":
        return "mad64_zero_addend" if re.search(r",\s*0\s*$", line.split(";")[0].rstrip()) else "mad64_register_addend"
    return "full_rate" if base in FULL else "half_rate"


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r02", "p2_issue_floor.json")
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "p2.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"),
                               "-I", os.path.join(ROOT, "hyperfridge-r0_amd", "csrc"), "--cuda-device-only", "-S", "-o", asm,
                               os.path.join(ROOT, "tools", "microbench", "p2_rounds_bench.hip")], stderr=subprocess.DEVNULL)
        lines = open(asm).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z13rounds_kernelILi2EE"))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end]
    # regions: [outer loop header .. end of outer loop]; inner loops (Depth=2 / "Parent Loop") run four times
    labels = [(i, l) for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)]
    outer = next(i for i, l in labels if "Loop Header" in l)
    inner = [i for i, l in labels if "Parent Loop" in l]
    assert len(inner) == 2, "expected two inner loops of full rounds"
    branches = [i for i, l in enumerate(body) if "s_cbranch" in l]
    inner_end = [next(b for b in branches if b > i) for i in inner]
    outer_end = next(b for b in branches if b > inner_end[1])
    weight = collections.Counter()
    classes = collections.Counter()
    for i in range(outer, outer_end):
        m = re.match(r"^\s+(v_[a-z0-9_]+)", body[i])
        if not m:
            continue
        times = 4 if any(a <= i < b for a, b in zip(inner, inner_end)) else 1
        weight[m.group(1)] += times
        classes[classify(m.group(1), body[i])] += times
    estimate = round(sum(RATE[c] * n for c, n in classes.items()))
    cycles = round(sum(FLOOR[c] * n for c, n in classes.items()))
    res = {"_about": "static count over tools/microbench/p2_rounds_bench.hip rounds_kernel<2> (same poseidon2_device.hpp as hash_rows / hash_fold)",
           "valu_instructions_per_permutation": sum(weight.values()), "by_class": dict(classes), "measured_simd_cycles_per_class": RATE, "floor_simd_cycles_per_class": FLOOR, "by_opcode": dict(weight.most_common()),
           "rate_table_estimate_simd_cycles_per_wave_permutation": estimate,
           "issue_floor_simd_cycles_per_wave_permutation": cycles}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: res[k] for k in ("valu_instructions_per_permutation", "by_class", "issue_floor_simd_cycles_per_wave_permutation",
                                          "rate_table_estimate_simd_cycles_per_wave_permutation")}))


if __name__ == "__main__":
    main()
