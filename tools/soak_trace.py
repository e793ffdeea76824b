#!/usr/bin/env python3
"""Soak run for the trace circuit on the GPU box: random RV32IM programs (ALU, M extension, loads / stores, forward branches of
every kind, forward JALs, a backward loop around everything, a READ_WORDS / COMMIT pair) are executed, the device expands their
compact preflight rows into the trace circuit's witness (r0h_trace_witgen) -- which must equal the host reference word for word --
and proves each; the seal must equal the CPU oracle's word for word and verify bound to the control root.  Every eighth run is also
tampered with (one register value read back wrong) and every other eighth carries a wrong result that keeps memory consistent (a dead
write): both verifiers must refuse both.
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
    """n random instructions of every RV32IM kind; branches and jumps only go forward (by 8 to 16 bytes), so the body terminates;
    x28 = scratch page, x29 = loop counter (the body runs 1-6 times), x30 = base of the JALRs."""
    body, last_jump = [], -9
    regs = [r for r in range(1, 28)]
    for _ in range(n):
        k = int(rng.integers(0, 10))
        rd, rs1, rs2 = (int(rng.choice(regs)) for _ in range(3))
        if k < 3:
            f3 = int(rng.integers(0, 8))
            body.append(R(int(rng.choice([0, 1, 0x20] if f3 in (0, 5) else [0, 1])), rs2, rs1, f3, rd))       # base ISA, SUB / SRA, M extension
        elif k < 5:
            f3 = int(rng.integers(0, 8))
            imm = int(rng.integers(0, 4096)) if f3 not in (1, 5) else int(rng.integers(0, 32)) | (0x400 if f3 == 5 and rng.random() < 0.5 else 0)
            body.append(I(imm, rs1, f3, rd, 0x13))                                                             # incl. SLLI / SRLI / SRAI
        elif k == 5:
            if rng.random() < 0.3 and len(body) - last_jump > 2:  # no forward jump may land on the JALR (x30 would be stale)
                last_jump = len(body) + 1
                body += [U(0, 30, 0x17), I(int(rng.choice([12, 13, 16, 17])), 30, 0, int(rng.choice([0, 1, 5])), 0x67)]  # auipc x30, 0; jalr rd, 12|16(x30), low bit dropped
            else:
                body.append(U(int(rng.integers(0, 1 << 20)), rd, int(rng.choice([0x37, 0x17]))))
        elif k == 6:
            f3 = int(rng.choice([0, 1, 2, 4, 5]))
            width = 1 if f3 in (0, 4) else 2 if f3 in (1, 5) else 4
            body.append(I(int(rng.integers(0, 1020 // width)) * width, 28, f3, rd, 0x03))                      # lb / lh / lw / lbu / lhu rd, off(x28)
        elif k == 7:
            f3 = int(rng.integers(0, 3))
            body.append(S(int(rng.integers(0, 1020 >> f3)) << f3, rs2, 28, f3))                                # sb / sh / sw rs2, off(x28)
        elif k == 8:
            last_jump = len(body)
            body.append(B(int(rng.choice([8, 12])), rs2, rs1, int(rng.choice([0, 1, 4, 5, 6, 7]))))
        else:
            last_jump = len(body)
            body.append(J(int(rng.choice([8, 12])), int(rng.choice([0, 1, 5]))))
    body += [ADDI(0, 0, 0)] * 3  # landing room for the last forward jumps
    loop = flat(body, ADDI(29, 29, -1), B(-4 * (len(body) + 1), 0, 29, 1))
    n_io = int(rng.integers(0, 9))
    jb = 0x20000000  # R0H_JOURNAL_BASE: COMMIT names words of the journal window, word i at jb + 4 i
    io = flat(LI(A0, jb), ADDI(11, 0, n_io), ADDI(A7, 0, 1), ECALL,                        # READ_WORDS(jb, n_io)
              LI(A0, jb), ADDI(11, 0, max(0, n_io - 1)), ADDI(A7, 0, 2), ECALL)               # COMMIT(jb, n_io - 1)
    return flat(LI(28, 0x40000), ADDI(29, 0, int(rng.integers(1, 7))), loop, io, ADDI(A0, 0, 0), ADDI(A7, 0, 0), ECALL)


def dead_write_lie(rows, bounds, rng):
    """A lie that keeps memory consistent: an instruction's result changed in a register nobody reads before it is written again (or
    never: then its boundary row's last value follows).  Only the constraints of the instruction's own unit can object.
    -> (rows, bounds) or None when the run has no such write."""
    F_INSN, F_RD, F_BEFORE, F_AFTER = 2, 6, 7, 8
    reads = lambda w: ((17, 10) if w[F_INSN] == 0x73 else ((int(w[F_INSN]) >> 15) & 31, (int(w[F_INSN]) >> 20) & 31))
    n = len(rows)
    for r in rng.permutation(n - 1)[:200]:
        w = rows[r]
        reg = int(w[F_RD])
        if not reg or w[F_INSN] == 0x73:
            continue
        nxt = next((q for q in range(r + 1, n) if reg in reads(rows[q]) or rows[q, F_RD] == reg), None)
        if nxt is not None and reg in reads(rows[nxt]):
            continue  # somebody looks at it
        bad_rows, bad_bounds = rows.copy(), bounds.copy()
        lie = int(w[F_AFTER]) ^ (1 << int(rng.integers(0, 32)))
        bad_rows[r, F_AFTER] = lie
        if nxt is not None:
            bad_rows[nxt, F_BEFORE] = lie           # the next write overwrites the lie
        else:
            k = int(np.nonzero(bounds[:, 0] == r0.REG_BASE + reg)[0][0])
            bad_bounds[k, 2] = lie                  # the value the segment leaves there
        return bad_rows, bad_bounds
    return None


def main():
    budget = float(sys.argv[1]) * 60 if len(sys.argv) > 1 else 120.0
    orc, hal = orc_binding.load(), r0.Hal(0)
    blob = np.fromfile(os.path.join(ROOT, "circuits", "trace.r0c"), dtype=np.uint32)
    oc, gc = orc.circuit(blob), hal.load_circuit(blob)
    rng = np.random.default_rng(2026)
    fixed = {}
    t0, n, rows_total, taken, tampered, lied, parity = time.time(), 0, 0, 0, 0, 0, 0
    while time.time() - t0 < budget:
        prog = random_program(rng, int(rng.integers(40, 990)))  # the closing branch reaches back at most 4 KiB
        vm = r0.Vm()
        vm.load(0x1000, prog)
        vm.set_pc(0x1000)
        for i in range(1, 28):
            vm.set_reg(i, int(rng.integers(0, 1 << 32)) if rng.random() < 0.7 else int(rng.choice([0, 1, 0xFFFFFFFF, 0x80000000])))
        vm.set_input([int(v) for v in rng.integers(0, 1 << 32, 8)])
        assert vm.run(segment_po2=20, keep_trace=True, boundary_rows=True) == (0, 0)
        rows, bounds = vm.preflight_arrays(0)
        po2 = max(r0.TRACE_MIN_PO2, int(np.ceil(np.log2(len(rows) + len(bounds)))))
        if po2 not in fixed:
            code, synthetic, _ = hal.witgen(gc, po2, 0)
            synthetic.free()
            fixed[po2] = (hal.code_commit(gc, po2, code), oc.witgen(po2, 0)[0], code)
        cc, ocode, code = fixed[po2]
        root = cc.root()
        data, glob = vm.trace_witness(0, po2)
        dev, dglob = hal.trace_witgen(rows, bounds, po2, circuit=gc)
        if not np.array_equal(dev.to_host(), data) or not np.array_equal(dglob, glob):
            print("FAILED on program %d: the device's witness differs from the host reference" % n)
            sys.exit(1)
        glob[r0.TRACE_GAMMA:r0.TRACE_GAMMA + 16] = [orc.enc(int(v)) for v in rng.integers(0, 2013265921, 16)]  # one seal outside a session: any challenge
        glob = hal.logup_totals(gc, po2, code, dev, glob)
        seal = hal.prove_segment(gc, po2, cc, dev, glob)
        if oc.verify(seal, code_root=root)[0] != 0 or r0.verify_seal(blob, seal, code_root=root)[0] != 0:
            print("FAILED on program %d (%d rows, po2 %d): a verifier refuses the device's seal" % (n, len(rows), po2))
            sys.exit(1)
        if n % 10 == 0:  # (the oracle takes seconds per 2^16-row proof: every tenth program is proved by it as well)
            if not np.array_equal(oc.logup_totals(po2, ocode, data, glob), glob) or not np.array_equal(seal, oc.prove(po2, ocode, data, glob)):
                print("FAILED on program %d (%d rows, po2 %d): the device's seal is not the oracle's" % (n, len(rows), po2))
                sys.exit(1)
            parity += 1
        if n % 8 == 0:  # a register read that does not return what was written: refused by both verifiers
            reads = [r for r in range(len(rows)) if (rows[r, 2] >> 15) & 31]
            bad = rows.copy()
            bad[reads[len(reads) // 2], 4] ^= 1 << int(rng.integers(0, 32))
            hal.trace_witgen(bad, bounds, po2, into=dev, circuit=gc)
            forged = hal.prove_segment(gc, po2, cc, dev, hal.logup_totals(gc, po2, code, dev, glob))
            if oc.verify(forged, code_root=root)[0] != 4 or r0.verify_seal(blob, forged, code_root=root)[0] != 4:
                print("FAILED on program %d: an inconsistent register read was accepted" % n)
                sys.exit(1)
            tampered += 1
        if n % 8 == 4:  # a register whose value changes between a dead write and the next write of it: refused by both verifiers
            lie = dead_write_lie(rows, bounds, rng)
            if lie is not None:
                hal.trace_witgen(lie[0], lie[1], po2, into=dev, circuit=gc)
                forged = hal.prove_segment(gc, po2, cc, dev, hal.logup_totals(gc, po2, code, dev, glob))
                if oc.verify(forged, code_root=root)[0] != 4 or r0.verify_seal(blob, forged, code_root=root)[0] != 4:
                    print("FAILED on program %d: a wrong result was accepted" % n)
                    sys.exit(1)
                lied += 1
        dev.free()
        n += 1
        rows_total += len(rows)
        taken += int((np.isin(rows[:, 2] & 0x7f, (0x63, 0x6f)) & (rows[:, 3] != rows[:, 1] + 4)).sum())
        if n % 50 == 0:
            print("%d executions proved (%d cycles, %d taken branches and jumps) after %.0f s" % (n, rows_total, taken, time.time() - t0), flush=True)
    print("soak ok: %d random executions (%d cycles, %d taken branches and jumps): device witness == host reference word for word, both verifiers accept every seal "
          "bound to the control root, %d seals proved by the oracle too and equal word for word; %d tampered runs (a register read back wrong) and %d runs with a "
          "dead register's value altered refused by both" % (n, rows_total, taken, parity, tampered, lied))


if __name__ == "__main__":
    main()
