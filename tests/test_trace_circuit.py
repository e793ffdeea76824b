"""The trace circuit (tools/gen_circuit.py trace, circuits/trace.r0c): a circuit whose DATA group IS the executor's preflight
trace (r0h_vm_trace_witness) -- the one place where what the prover commits to comes from an execution rather than from a
synthetic column program.  It constrains that the rows form one contiguous run from the public
first pc to the public last pc in the public number of cycles, and that control flow follows the instruction words (a step
leaves pc + 4 only at JAL / JALR / branch words; JAL and branches go where their immediates say).  Register and memory contents,
branch conditions and JALR targets are risc0's rv32im circuit's business (its tap table and constraint polynomial cannot be
reproduced here) and are not constrained.  The non-gpu tests prove with the oracle; both verifiers check."""
import numpy as np
import pytest

import hyperfridge_r0_amd as r0
from conftest import circuit_path
from test_rv32im import _guest

import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
from gen_circuit import TRACE_COLUMNS  # noqa: E402

COL = {name: i for i, name in enumerate(TRACE_COLUMNS)}
P = 2013265921


def _run(n_loop=60):
    prog, base = _guest(n_loop), 0x400
    vm = r0.Vm()
    vm.load(base, prog)
    vm.set_pc(base)
    vm.set_input([7, 0x01020304])
    assert vm.run(segment_po2=20, keep_trace=True) == (0, 0)
    return vm, base


def test_the_witness_is_the_preflight_trace(orc):
    vm, base = _run()
    rows = vm.preflight(0)
    n, po2 = len(rows), 10
    assert 256 < n <= 1 << po2
    data, glob = vm.trace_witness(0, po2)
    m = np.array([orc.dec(int(w)) for w in data], dtype=np.uint64).reshape(r0.TRACE_COLUMNS, 1 << po2)
    assert [orc.dec(int(g)) for g in glob] == [base, rows[-1].next_pc, n]
    assert m[COL["live"]].tolist() == [1] * n + [0] * ((1 << po2) - n)
    assert not m[:COL["inv_jal"], n:].any()  # blank past the end, but for the inverses that pin opcode 0 to "no jump"
    for k, code in enumerate((0x6F, 0x67, 0x63)):
        assert (m[COL["inv_jal"] + k, n] * (P - code)) % P == 1
    for r in (0, 1, n // 2, n - 1):
        w = rows[r]
        assert (m[COL["pc"], r], m[COL["next_pc"], r], m[COL["cycle"], r]) == (w.pc, w.next_pc, r)
        assert m[COL["insn_lo"], r] | m[COL["insn_hi"], r] << 16 == w.insn and m[COL["rd_after_lo"], r] | m[COL["rd_after_hi"], r] << 16 == w.rd_after
        assert m[COL["mem_kind"], r] == w.mem_kind and m[COL["mem_after_lo"], r] | m[COL["mem_after_hi"], r] << 16 == w.mem_after
    with pytest.raises(r0.R0HipError, match="do not fit"):
        vm.trace_witness(0, 8)


def test_an_execution_proves_and_an_altered_one_does_not(orc):
    vm, base = _run()
    n, po2 = len(vm.preflight(0)), 10
    N = 1 << po2
    data, glob = vm.trace_witness(0, po2)
    blob = np.fromfile(circuit_path("trace"), dtype=np.uint32)
    c = orc.circuit(blob)
    code, _, _ = c.witgen(po2, 0)  # the fixed CODE columns (first / last row, row index): the program's control root comes from them
    root = c.code_root(code, po2)
    seal = c.prove(po2, code, data, glob)
    assert c.verify(seal, code_root=root) == (0, "ok")
    assert r0.verify_seal(blob, seal, code_root=root)[:2] == (0, "ok")
    enc = orc.enc

    def rejected(d, g):
        s = c.prove(po2, code, d, g)
        got = c.verify(s, code_root=root)
        assert got == r0.verify_seal(blob, s, code_root=root)[:2]
        return got[0] == 4  # the constraint identity at z fails

    def edit(col, row, value):
        d = data.copy()
        d[COL[col] * N + row] = enc(value)
        return d

    mid = n // 2
    assert rejected(edit("pc", mid, 0x5000), glob)                    # a row that starts somewhere its predecessor did not go
    assert rejected(edit("next_pc", mid, 0x5000), glob)               # ... or goes somewhere the next one does not start
    assert rejected(edit("cycle", mid, mid + 1), glob)                # a skipped cycle
    assert rejected(edit("live", mid, 0), glob)                       # a hole in the run
    assert rejected(edit("live", n, 1), glob)                         # a row smuggled in after the end
    assert rejected(edit("mem_kind", mid, 3), glob)                   # not none / read / write
    rd = next(r for r, w in enumerate(vm.preflight(0)) if w.mem_kind == r0.MEM_READ)
    assert rejected(edit("mem_after_lo", rd, (vm.preflight(0)[rd].mem_after & 0xffff) ^ 1), glob)  # a read that changes the word
    for k, wrong in ((0, base + 4), (1, 0x5000), (2, n - 1)):         # public inputs that do not describe this run
        g = glob.copy()
        g[k] = enc(wrong)
        assert rejected(data, g)
    # control flow follows the instruction words
    rows = vm.preflight(0)
    br = next(r for r, w in enumerate(rows) if (w.insn & 0x7f) == 0x63 and w.next_pc != w.pc + 4)   # a taken branch
    assert rejected(edit("is_branch", br, 0), glob)                   # ... cannot pass as an ordinary instruction
    assert rejected(edit("bit0", br, 0), glob)                        # ... nor can its word be changed under it (halves and opcode pin the bits)
    d2 = edit("next_pc", br, rows[br].pc + 8)                         # ... nor can it go anywhere but pc + 4 or pc + imm_B,
    d2[COL["pc"] * N + br + 1] = enc(rows[br].pc + 8)                 #     even if the next row plays along
    assert rejected(d2, glob)
    alu = next(r for r, w in enumerate(rows) if (w.insn & 0x7f) == 0x13 and r > 4)
    d3 = edit("next_pc", alu, rows[alu].pc + 8)                       # an ALU instruction that skips the next one
    d3[COL["is_seq"] * N + alu] = enc(0)
    d3[COL["pc"] * N + alu + 1] = enc(rows[alu].pc + 8)
    assert rejected(d3, glob)
    assert rejected(edit("is_jal", alu, 1), glob)                     # ... and cannot be flagged as a jump to get away with it
    # what the circuit does not see, by design: what an instruction computes
    assert not rejected(edit("rd_after_lo", alu, 0x1234), glob)


def test_jumps_and_branches_of_every_kind_satisfy_the_control_flow_constraints(orc):
    """JAL forwards and backwards, JALR, taken and untaken branches with positive and negative offsets (the loads / stores /
    branches / jumps program of test_rv32im): the circuit accepts the genuine trace -- immediates are decoded as the ISA
    encodes them."""
    from test_rv32im import A0, A7, ADDI, ECALL, J, B, I, flat, T0, T1
    prog = flat(ADDI(T0, 0, 3),
                J(12, 1),                      # jal ra, +12  (skips two)
                ADDI(T1, T1, 100), ADDI(T1, T1, 100),
                ADDI(T0, T0, -1),              # loop:
                B(8, 0, T0, 0),                # beq t0, x0, +8 -> out
                J(-8, 0),                      # jal x0, loop
                I(0, 1, 0, 5, 0x67),           # out: jalr t0, ra, 0 -> back to the two skipped instructions
                ADDI(A0, 0, 0), ADDI(A7, 0, 0), ECALL)
    vm = r0.Vm()
    vm.load(0x1000, prog)
    vm.set_pc(0x1000)
    try:
        vm.run(segment_po2=20, keep_trace=True, max_cycles=200)
    except r0.R0HipError:
        pass  # wherever it ends, the rows so far are a run
    rows = vm.preflight(0)
    kinds = {w.insn & 0x7f for w in rows}
    assert {0x6f, 0x67, 0x63} <= kinds and any(w.next_pc < w.pc for w in rows)
    po2 = 9
    data, glob = vm.trace_witness(0, po2)
    blob = np.fromfile(circuit_path("trace"), dtype=np.uint32)
    c = orc.circuit(blob)
    code, _, _ = c.witgen(po2, 0)
    seal = c.prove(po2, code, data, glob)
    assert c.verify(seal, code_root=c.code_root(code, po2)) == (0, "ok")


@pytest.mark.gpu
@pytest.mark.parametrize("n_loop,po2", [(60, 10), (9000, 16)])
def test_the_device_proves_an_execution_trace_word_for_word_like_the_cpu_port(hal, orc, n_loop, po2):
    """The same on the GPU: CODE columns generated on the device, DATA uploaded from the executor's trace, the seal equal to the
    CPU port's and accepted by both verifiers bound to the control root; a row that breaks the run is rejected."""
    vm, base = _run(n_loop)
    n = len(vm.preflight(0))
    assert (1 << (po2 - 2)) < n <= (1 << po2)
    data, glob = vm.trace_witness(0, po2)
    blob = np.fromfile(circuit_path("trace"), dtype=np.uint32)
    c = orc.circuit(blob)
    gc = hal.load_circuit(blob)  # eval_check compiled in-process (hipRTC)
    code, synthetic, _ = hal.witgen(gc, po2, 0)
    synthetic.free()
    dev = hal.copy_from(data)
    seal = hal.prove_segment(gc, po2, code, dev, glob)
    root = hal.code_root(gc, po2, code)
    assert c.verify(seal, code_root=root) == (0, "ok") and r0.verify_seal(blob, seal, code_root=root)[:2] == (0, "ok")
    ocode, _, _ = c.witgen(po2, 0)
    assert np.array_equal(ocode, code.to_host())
    assert np.array_equal(seal, c.prove(po2, ocode, data, glob))
    bad = data.copy()
    bad[COL["pc"] * (1 << po2) + n // 2] = orc.enc(0x5000)
    dev.upload(bad)
    seal = hal.prove_segment(gc, po2, code, dev, glob)
    assert c.verify(seal, code_root=root)[0] == 4 and r0.verify_seal(blob, seal, code_root=root)[0] == 4
    code.free(); dev.free(); gc.free()
