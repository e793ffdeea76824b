#!/usr/bin/env python3
"""Soak run for the trace circuit on the GPU box: random RV32IM programs (ALU, M extension, loads / stores, forward branches of
every kind, forward JALs, a backward loop around everything) are executed, their preflight traces become witnesses, the device
proves each and the seal must equal the CPU oracle's word for word and verify bound to the control root.
usage: python tools/soak_trace.py [minutes]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import hyperfridge_r0_amd as r0
import orc_binding
from test_rv32im import ADDI, A0, A7, B, ECALL, I, J, LI, R, S, U, flat


def random_program(rng, n):
    """n random instructions; branches and jumps only go forward (by 8 or 12 bytes), so the body terminates; x28 = scratch page,
    x29 = loop counter (the body runs 1-6 times)."""
    body = []
    regs = [r for r in range(1, 28)]
    for _ in range(n):
        k = int(rng.integers(0, 10))
        rd, rs1, rs2 = (int(rng.choice(regs)) for _ in range(3))
        if k < 3:
            body.append(R(int(rng.choice([0, 1])), rs2, rs1, int(rng.integers(0, 8)), rd))
        elif k < 5:
            body.append(I(int(rng.integers(0, 4096)), rs1, int(rng.choice([0, 2, 3, 4, 6, 7])), rd, 0x13))
        elif k == 5:
            body.append(U(int(rng.integers(0, 1 << 20)), rd, int(rng.choice([0x37, 0x17]))))
        elif k == 6:
            body.append(I(int(rng.integers(0, 255)) * 4, 28, 2, rd, 0x03))           # lw rd, off(x28)
        elif k == 7:
            body.append(S(int(rng.integers(0, 255)) * 4, rs2, 28, 2))                # sw rs2, off(x28)
        elif k == 8:
            body.append(B(int(rng.choice([8, 12])), rs2, rs1, int(rng.choice([0, 1, 4, 5, 6, 7]))))
        else:
            body.append(J(int(rng.choice([8, 12])), int(rng.choice([0, 1, 5]))))
    body += [ADDI(0, 0, 0)] * 3  # landing room for the last forward jumps
    loop = flat(body, ADDI(29, 29, -1), B(-4 * (len(body) + 1), 0, 29, 1))
    return flat(LI(28, 0x40000), ADDI(29, 0, int(rng.integers(1, 7))), loop, ADDI(A0, 0, 0), ADDI(A7, 0, 0), ECALL)


def main():
    budget = float(sys.argv[1]) * 60 if len(sys.argv) > 1 else 120.0
    orc, hal = orc_binding.load(), r0.Hal(0)
    blob = np.fromfile(os.path.join(ROOT, "circuits", "trace.r0c"), dtype=np.uint32)
    oc, gc = orc.circuit(blob), hal.load_circuit(blob)
    rng = np.random.default_rng(2026)
    fixed = {}
    t0, n, rows_total, taken = time.time(), 0, 0, 0
    while time.time() - t0 < budget:
        prog = random_program(rng, int(rng.integers(40, 990)))  # the closing branch reaches back at most 4 KiB
        vm = r0.Vm()
        vm.load(0x1000, prog)
        vm.set_pc(0x1000)
        for i in range(1, 28):
            vm.set_reg(i, int(rng.integers(0, 1 << 32)) if rng.random() < 0.7 else int(rng.choice([0, 1, 0xFFFFFFFF, 0x80000000])))
        assert vm.run(segment_po2=20, keep_trace=True) == (0, 0)
        rows = vm.preflight(0)
        po2 = max(9, int(np.ceil(np.log2(len(rows)))))
        if po2 not in fixed:
            code, synthetic, _ = hal.witgen(gc, po2, 0)
            synthetic.free()
            fixed[po2] = (code, hal.code_root(gc, po2, code), oc.witgen(po2, 0)[0])
        code, root, ocode = fixed[po2]
        data, glob = vm.trace_witness(0, po2)
        dev = hal.copy_from(data)
        seal = hal.prove_segment(gc, po2, code, dev, glob)
        dev.free()
        want = oc.prove(po2, ocode, data, glob)
        if not np.array_equal(seal, want) or oc.verify(seal, code_root=root)[0] != 0 or r0.verify_seal(blob, seal, code_root=root)[0] != 0:
            print("FAILED on program %d (%d rows, po2 %d)" % (n, len(rows), po2))
            sys.exit(1)
        n += 1
        rows_total += len(rows)
        taken += sum(1 for w in rows if (w.insn & 0x7f) in (0x63, 0x6f) and w.next_pc != w.pc + 4)
        if n % 50 == 0:
            print("%d executions proved (%d cycles, %d taken branches and jumps) after %.0f s" % (n, rows_total, taken, time.time() - t0), flush=True)
    print("soak ok: %d random executions (%d cycles, %d taken branches and jumps): device seal == oracle seal word for word, both verifiers accept bound to the control root"
          % (n, rows_total, taken))


if __name__ == "__main__":
    main()
