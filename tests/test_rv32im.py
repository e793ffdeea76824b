"""RV32IM executor, segmenter and preflight trace (csrc/rv32im.cpp).  The reference ships no guest ELF
(only sources: methods/guest/src/main.rs), so programs are hand-encoded instruction words.  Instruction semantics are checked
against an independent interpreter written here in Python from the RISC-V specification; the ecall ABI, the cycle model and the
page Merkle root are this library's own (documented in the source) and are checked for self-consistency: segments chain, the
claims they yield verify as a receipt's claims do, the trace replays."""
import hashlib
import os
import struct
import sys

import numpy as np
import pytest

import hyperfridge_r0_amd as r0

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

M32 = 0xFFFFFFFF


# ---- a tiny assembler (encodings from the RISC-V unprivileged specification, chapter 2 and the M extension)
def R(f7, rs2, rs1, f3, rd, op=0x33):
    return (f7 << 25) | (rs2 << 20) | (rs1 << 15) | (f3 << 12) | (rd << 7) | op


def I(imm, rs1, f3, rd, op):
    return ((imm & 0xFFF) << 20) | (rs1 << 15) | (f3 << 12) | (rd << 7) | op


def S(imm, rs2, rs1, f3):
    return (((imm >> 5) & 0x7F) << 25) | (rs2 << 20) | (rs1 << 15) | (f3 << 12) | ((imm & 0x1F) << 7) | 0x23


def B(imm, rs2, rs1, f3):
    return (((imm >> 12) & 1) << 31) | (((imm >> 5) & 0x3F) << 25) | (rs2 << 20) | (rs1 << 15) | (f3 << 12) | (((imm >> 1) & 0xF) << 8) | (((imm >> 11) & 1) << 7) | 0x63


def U(imm20, rd, op):
    return ((imm20 & 0xFFFFF) << 12) | (rd << 7) | op


def J(imm, rd):
    return (((imm >> 20) & 1) << 31) | (((imm >> 1) & 0x3FF) << 21) | (((imm >> 11) & 1) << 20) | (((imm >> 12) & 0xFF) << 12) | (rd << 7) | 0x6F


ADDI = lambda rd, rs1, imm: I(imm, rs1, 0, rd, 0x13)
LI = lambda rd, v: [U(((v + 0x800) >> 12) & 0xFFFFF, rd, 0x37), ADDI(rd, rd, v & 0xFFF)]  # lui + addi (sign-extension aware)
ECALL = 0x73
A0, A1, A7, T0, T1, T2, S0, S1 = 10, 11, 17, 5, 6, 7, 8, 9


def flat(*parts):
    out = []
    for p in parts:
        out.extend(p if isinstance(p, list) else [p])
    return out


# ---- the independent interpreter: straight from the specification's tables, Python integers
def sx(v, bits):
    v &= (1 << bits) - 1
    return v - (1 << bits) if v >> (bits - 1) else v


def py_step(x, pc, mem, insn):
    """Executes one non-ecall instruction on register list x / byte-addressed dict mem; returns the next pc."""
    op, rd, f3, rs1, rs2, f7 = insn & 0x7F, (insn >> 7) & 31, (insn >> 12) & 7, (insn >> 15) & 31, (insn >> 20) & 31, insn >> 25
    a, b = x[rs1], x[rs2]
    imm_i = sx(insn >> 20, 12)
    nxt, val = pc + 4, None
    if op == 0x37:
        val = insn & 0xFFFFF000
    elif op == 0x17:
        val = pc + (insn & 0xFFFFF000)
    elif op == 0x6F:
        imm = sx(((insn >> 31) << 20) | (((insn >> 12) & 0xFF) << 12) | (((insn >> 20) & 1) << 11) | (((insn >> 21) & 0x3FF) << 1), 21)
        val, nxt = pc + 4, pc + imm
    elif op == 0x67:
        val, nxt = pc + 4, (a + imm_i) & ~1
    elif op == 0x63:
        imm = sx(((insn >> 31) << 12) | (((insn >> 7) & 1) << 11) | (((insn >> 25) & 0x3F) << 5) | (((insn >> 8) & 0xF) << 1), 13)
        sa, sb = sx(a, 32), sx(b, 32)
        take = {0: a == b, 1: a != b, 4: sa < sb, 5: sa >= sb, 6: a < b, 7: a >= b}[f3]
        if take:
            nxt = pc + imm
    elif op == 0x03:
        addr = (a + imm_i) & M32
        n = {0: 1, 1: 2, 2: 4, 4: 1, 5: 2}[f3]
        raw = sum(mem.get(addr + i, 0) << (8 * i) for i in range(n))
        val = sx(raw, 8 * n) if f3 in (0, 1) else raw
    elif op == 0x23:
        imm = sx(((insn >> 25) << 5) | ((insn >> 7) & 31), 12)
        addr = (a + imm) & M32
        for i in range(1 << f3):
            mem[addr + i] = (b >> (8 * i)) & 0xFF
    elif op == 0x13:
        sh = rs2
        val = {0: a + imm_i, 2: int(sx(a, 32) < imm_i), 3: int(a < (imm_i & M32)), 4: a ^ (imm_i & M32), 6: a | (imm_i & M32), 7: a & (imm_i & M32),
               1: a << sh, 5: (sx(a, 32) >> sh) if f7 == 0x20 else (a >> sh)}[f3]
    elif op == 0x33:
        sa, sb = sx(a, 32), sx(b, 32)
        if f7 == 1:
            def div(p, q):  # round toward zero
                return abs(p) // abs(q) * (1 if (p < 0) == (q < 0) else -1)
            val = {0: a * b, 1: (sa * sb) >> 32, 2: (sa * b) >> 32, 3: (a * b) >> 32,
                   4: -1 if b == 0 else (sa if (sa == -2**31 and sb == -1) else div(sa, sb)),
                   5: M32 if b == 0 else a // b,
                   6: sa if b == 0 else (0 if (sa == -2**31 and sb == -1) else sa - sb * div(sa, sb)),
                   7: a if b == 0 else a % b}[f3]
        else:
            alt = f7 == 0x20
            val = {0: a - b if alt else a + b, 1: a << (b & 31), 2: int(sa < sb), 3: int(a < b), 4: a ^ b, 5: (sa >> (b & 31)) if alt else (a >> (b & 31)), 6: a | b, 7: a & b}[f3]
    elif op == 0x0F:
        pass
    else:
        raise ValueError("py_step: opcode %#x" % op)
    if val is not None and rd:
        x[rd] = val & M32
    return nxt & M32


def run_py(words, base, regs, max_steps=100000):
    mem = {}
    for i, w in enumerate(words):
        for k in range(4):
            mem[base + 4 * i + k] = (w >> (8 * k)) & 0xFF
    x, pc, steps = list(regs), base, 0
    while True:
        insn = sum(mem.get(pc + k, 0) << (8 * k) for k in range(4))
        if insn == ECALL:
            return x, pc, steps, mem
        pc = py_step(x, pc, mem, insn)
        steps += 1
        assert steps < max_steps


def test_random_alu_and_m_extension_programs_match_the_specification():
    rng = np.random.default_rng(11)
    specials = [0, 1, 2, 0x7FFFFFFF, 0x80000000, 0xFFFFFFFF, 0xFFFFFFFE, 31, 32, 0x12345678]
    for trial in range(60):
        regs = [0] + [int(rng.choice(specials)) if rng.random() < 0.5 else int(rng.integers(0, 1 << 32)) for _ in range(31)]
        prog = []
        for _ in range(120):
            kind = int(rng.integers(0, 4))
            rd, rs1, rs2 = (int(v) for v in rng.integers(0, 32, 3))
            if kind == 0:
                prog.append(R(1, rs2, rs1, int(rng.integers(0, 8)), rd))               # M extension
            elif kind == 1:
                f3 = int(rng.integers(0, 8))
                prog.append(R(0x20 if f3 in (0, 5) and rng.random() < 0.5 else 0, rs2, rs1, f3, rd))
            elif kind == 2:
                f3 = int(rng.choice([0, 2, 3, 4, 6, 7]))
                prog.append(I(int(rng.integers(0, 4096)), rs1, f3, rd, 0x13))
            else:
                f3 = int(rng.choice([1, 5]))
                prog.append(I((0x400 if f3 == 5 and rng.random() < 0.5 else 0) | int(rng.integers(0, 32)), rs1, f3, rd, 0x13))
        prog += [U(int(rng.integers(0, 1 << 20)), 3, 0x37), U(int(rng.integers(0, 1 << 20)), 4, 0x17)]
        prog.append(ECALL)  # a7 is whatever the program left: force HALT below by patching a7 just before
        prog[-1:] = [ADDI(A7, 0, 0), ECALL]
        base = 0x1000
        want, _, steps, _ = run_py(prog, base, regs)
        vm = r0.Vm()
        vm.load(base, prog)
        vm.set_pc(base)
        for i in range(1, 32):
            vm.set_reg(i, regs[i])
        kind, code = vm.run()
        assert kind == r0.Vm.HALTED and vm.cycles == steps + 1
        got = [vm.reg(i) for i in range(32)]
        assert got == want, (trial, [(i, hex(g), hex(w)) for i, (g, w) in enumerate(zip(got, want)) if g != w])
        assert code == want[A0]


def test_m_extension_corner_cases_from_the_specification_table():
    # (op f3, rs1, rs2) -> result: division by zero and signed overflow, RISC-V spec table 7.1
    cases = [(4, 7, 0, M32), (5, 7, 0, M32), (6, 7, 0, 7), (7, 7, 0, 7), (4, 0x80000000, M32, 0x80000000), (6, 0x80000000, M32, 0),
             (4, (-7) & M32, 2, (-3) & M32), (6, (-7) & M32, 2, (-1) & M32), (1, 0x80000000, 0x80000000, 0x40000000), (2, M32, M32, M32), (3, M32, M32, 0xFFFFFFFE)]
    for f3, a, b, want in cases:
        vm = r0.Vm()
        vm.load(0, [R(1, 6, 5, f3, 7), ADDI(A7, 0, 0), ECALL])
        vm.set_reg(5, a)
        vm.set_reg(6, b)
        vm.run()
        assert vm.reg(7) == want, (f3, hex(a), hex(b), hex(vm.reg(7)))


def test_loads_stores_branches_and_jumps():
    base, data = 0x2000, 0x8000
    prog = flat(LI(S0, data), LI(T0, 0x80C3F1A5),
                S(0, T0, S0, 2),                      # sw
                I(0, S0, 0, 12, 0x03), I(1, S0, 4, 13, 0x03), I(2, S0, 1, 14, 0x03), I(2, S0, 5, 15, 0x03), I(3, S0, 0, 16, 0x03),  # lb lbu lh lhu lb
                S(5, T0, S0, 0), S(6, T0, S0, 1),     # sb at +5, sh at +6
                I(4, S0, 2, 18, 0x03),                # lw +4
                # loop: t1 = sum 1..10
                ADDI(T1, 0, 0), ADDI(T2, 0, 10),
                R(0, T2, T1, 0, T1), ADDI(T2, T2, -1), B(-8, 0, T2, 1),   # add; addi; bne t2, x0, -8
                J(8, 1), ADDI(T1, T1, 1000),          # jal skips the addi
                ADDI(19, 1, 0),                       # x19 = return address left by jal
                ADDI(A0, T1, 0), ADDI(A7, 0, 0), ECALL)
    want, _, steps, mem = run_py(prog, base, [0] * 32)
    vm = r0.Vm()
    vm.load(base, prog)
    vm.set_pc(base)
    kind, code = vm.run(keep_trace=True)
    assert kind == 0 and code == 55 and [vm.reg(i) for i in range(32)] == want
    assert vm.reg(12) == 0xFFFFFFA5 and vm.reg(13) == 0xF1 and vm.reg(14) == 0xFFFF80C3 and vm.reg(15) == 0x80C3 and vm.reg(16) == 0xFFFFFF80
    assert vm.reg(18) == 0xF1A5A500 and vm.read(data, 2).tolist() == [0x80C3F1A5, 0xF1A5A500]
    # the preflight trace replays: registers and memory rebuilt from the rows alone
    rows = vm.preflight(0)
    assert len(rows) == steps + 1 and [r.cycle for r in rows] == list(range(len(rows)))
    x, memw = [0] * 32, {}
    for k, r in enumerate(rows):
        assert r.pc == (rows[k - 1].next_pc if k else base)
        if r.mem_kind == 2:
            assert memw.get(r.mem_addr, 0) == r.mem_before
            memw[r.mem_addr] = r.mem_after
        elif r.mem_kind == 1:
            assert memw.get(r.mem_addr, 0) == r.mem_before == r.mem_after
        if r.rd:
            x[r.rd] = r.rd_after
    assert x == want and memw == {data: 0x80C3F1A5, data + 4: 0xF1A5A500}
    # traps are errors, as a guest panic is an Err from `prove`
    for bad, why in [([0xFFFFFFFF], "illegal"), ([I(2, 0, 2, 5, 0x03)], "misaligned load"), (flat(ADDI(A7, 0, 99), ECALL), "unknown ecall"), ([0x00100073], "ebreak")]:
        vm = r0.Vm()
        vm.load(0, bad)
        with pytest.raises(r0.R0HipError, match=why):
            vm.run()


def _guest(n_loop):
    """reads two input words, loops n_loop times doing memory traffic over several pages, commits two words, halts with 0"""
    buf, scratch = r0.JOURNAL_BASE, 0x20000  # (the buffer is the head of the journal window: COMMIT names words there)
    return flat(LI(A0, buf), ADDI(A1, 0, 2), ADDI(A7, 0, 1), ECALL,              # READ_WORDS(buf, 2)
                LI(S0, scratch), LI(T2, n_loop), ADDI(T1, 0, 0),
                # loop body: store counter at scratch + (t1 & 0xFFC) * 16 (walks pages), accumulate
                I(0xFFC, T1, 7, T0, 0x13), I(4, T0, 1, T0, 0x13), R(0, S0, T0, 0, T0), S(0, T1, T0, 2), ADDI(T1, T1, 4), ADDI(T2, T2, -1), B(-24, 0, T2, 1),
                LI(A0, buf), I(0, A0, 2, T0, 0x03), R(0, T1, T0, 0, T0), S(0, T0, A0, 2),   # buf[0] += t1
                ADDI(A1, 0, 2), ADDI(A7, 0, 2), ECALL,                                      # COMMIT(buf, 2 words)
                ADDI(A7, 0, 3), ECALL, ADDI(S1, A0, 0),                                     # CYCLES -> s1
                ADDI(A0, 0, 0), ADDI(A7, 0, 0), ECALL)


def test_segmenter_cuts_a_run_and_the_claims_chain_like_a_receipts(orc):
    prog, base = _guest(3000), 0x400
    def run(po2, **kw):
        vm = r0.Vm()
        vm.load(base, prog)
        vm.set_pc(base)
        vm.set_input([7, 0x01020304])
        return vm, vm.run(segment_po2=po2, **kw)
    one, (kind, code) = run(20)
    assert (kind, code) == (0, 0) and len(one.segments()) == 1
    total = one.cycles
    assert one.journal == struct.pack("<II", 7 + 4 * 3000, 0x01020304) and one.reg(S1) == total - 5  # CYCLES counts the instructions before itself; four more and the ecall follow
    vm, (kind, code) = run(12, page_in_cycles=20, page_out_cycles=30)
    segs = vm.segments()
    assert (kind, code) == (0, 0) and vm.journal == one.journal and vm.cycles == total and len(segs) >= 6
    assert sum(s.user_cycles for s in segs) == total and [s.index for s in segs] == list(range(len(segs)))
    for k, s in enumerate(segs):
        assert s.user_cycles + s.paging_cycles <= 1 << 12 and s.paging_cycles == 20 * s.pages_in + 30 * s.pages_out and s.pages_in >= 1
        last = k == len(segs) - 1
        assert (s.exit_system, s.exit_user) == ((0, 0) if last else (2, 0))
        if not last:
            assert bytes(s.post.merkle_root) == bytes(segs[k + 1].pre.merkle_root) and s.post.pc == segs[k + 1].pre.pc
            assert bytes(s.post.merkle_root) != bytes(s.pre.merkle_root)
    # the same program, the same image id; another program, another one
    assert segs[0].pre.digest() == one.segments()[0].pre.digest()
    other = r0.Vm()
    other.load(base, prog[:-1] + [ADDI(0, 0, 0)])
    other.set_pc(base)
    assert other.run(max_cycles=1)[0] == r0.Vm.LIMIT and other.segments()[0].pre.digest() != segs[0].pre.digest()
    assert (other.segments()[-1].exit_system, other.segments()[-1].exit_user) == (2, 2)  # SessionLimit
    # the page Merkle root, recomputed independently for a one-page image
    tiny = r0.Vm()
    tiny.load(0x800, [ADDI(A7, 0, 0), ECALL])
    tiny.set_pc(0x800)
    tiny.run()
    page = bytearray(1024)
    page[0:8] = struct.pack("<II", ADDI(A7, 0, 0), ECALL)
    h = hashlib.sha256(bytes(page)).digest()
    z = hashlib.sha256(bytes(1024)).digest()
    idx = 0x800 >> 10
    for level in range(22):
        h = hashlib.sha256((z + h) if (idx >> level) & 1 else (h + z)).digest()
        z = hashlib.sha256(z + z).digest()
    assert bytes(tiny.segments()[0].post.merkle_root) == h  # (the run stores nothing: the state it leaves has the image's pages)
    # ... and the root of the state a run STARTS from -- what the image id names -- is the Poseidon2 digest of the image's word list
    # (tools/image_circuit.py: the form the image circuit can tie to the session's memory argument), restated there in Python
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import image_circuit
    _, digest = image_circuit.witness([(0x800 // 4, ADDI(A7, 0, 0)), (0x800 // 4 + 1, ECALL)], 64)
    assert bytes(tiny.segments()[0].pre.merkle_root) == b"".join(int(w).to_bytes(4, "little") for w in digest)
    # the claims of the run are what a composite receipt carries: prove each segment for its claim (CPU oracle) and verify the receipt
    claims = vm.claims()
    assert claims[-1].digest() != claims[0].digest() and bytes(claims[-1].output_digest) == r0.output_digest(vm.journal)
    from conftest import circuit_path
    blob = np.fromfile(circuit_path("small"), dtype=np.uint32)
    c = orc.circuit(blob)
    seals = []
    for k, cl in enumerate(claims):
        code, data, glob = c.witgen(9, 500 + k, globals_in=cl.globals())
        seals.append(c.prove(9, code, data, glob))
    roots = {9: c.code_root(code, 9)}
    rc = r0.Receipt.new(vm.journal, seals, claims)
    assert rc.verify(blob, roots, segs[0].pre.digest())[:2] == (0, "ok")
    assert rc.verify(blob, roots, other.segments()[0].pre.digest())[0] == 8  # another program's image id


def test_an_io_ecall_moves_one_word_per_cycle_and_can_be_cut_anywhere():
    """tools/fuzz (round 2) found that a READ_WORDS or COMMIT which touches every page of its buffer inside ONE cycle cannot be
    priced like an ordinary instruction.  Since round 3 the two I/O ecalls re-execute once per word: while a1 = j > 0 the cycle
    moves word j - 1 of the buffer and writes a1 = j - 1, with a1 = 0 the ecall falls through (n + 1 cycles for n words, no state
    outside the registers -- what the trace circuit constrains).  A cycle has one memory access, pays for at most its own pages,
    and a transfer of any length is cut between two segments like any other stretch of the run; the journal and the memory come
    out in stream order.  Round 4: the journal is a window of memory -- COMMIT names the words at R0H_JOURNAL_BASE + 4 i, each once."""
    buf = r0.JOURNAL_BASE - 4  # (so that buf + 4, the source of the COMMIT below, is the head of the journal window)
    def prog(n_words):
        return flat([ADDI(T0, T0, 1)] * 40, LI(A0, buf), LI(A1, n_words), ADDI(A7, 0, 1), ECALL,   # 40 cheap cycles, then READ_WORDS(buf, n)
                    LI(A0, buf + 4), LI(A1, n_words - 2), ADDI(A7, 0, 2), ECALL,                     # COMMIT(buf + 4, n - 2 words)
                    ADDI(A0, 0, 0), ADDI(A7, 0, 0), ECALL)
    def run(n_words, po2, cin, cout, **kw):
        vm = r0.Vm()
        vm.load(0x400, prog(n_words))
        vm.set_pc(0x400)
        vm.set_input(list(range(1, n_words + 1)))
        return vm, vm.run(segment_po2=po2, page_in_cycles=cin, page_out_cycles=cout, **kw)
    n = 20 * 256  # 20 pages of 1 KiB
    want = struct.pack("<%dI" % n, *range(1, n + 1))[4:-4]
    for cin, cout in ((8, 8), (30, 30)):
        vm, (kind, code) = run(n, 9, cin, cout)
        segs = vm.segments()
        assert (kind, code) == (0, 0) and len(segs) >= 20 and vm.journal == want and vm.read(buf, n).tolist() == list(range(1, n + 1))
        assert vm.reg(A1) == 0 and vm.cycles == 40 + 2 + 2 + 1 + (n + 1) + 2 + 2 + 1 + (n - 2 + 1) + 3  # one cycle per word moved and one to fall through
        for s in segs:
            assert s.user_cycles + s.paging_cycles <= 1 << 9, (s.index, s.user_cycles, s.paging_cycles)
    # the rows of such a transfer: the ecall word at one pc, one memory word each, a1 written every time
    vm, _ = run(6, 20, 0, 0, keep_trace=True)
    io = [w for w in vm.preflight(0) if w.insn == ECALL and w.rs1_value == 1]
    assert len(io) == 7 and len({w.pc for w in io}) == 1 and [w.next_pc - w.pc for w in io] == [0] * 6 + [4]
    assert [w.mem_kind for w in io] == [r0.MEM_WRITE] * 6 + [r0.MEM_NONE] and [w.mem_addr for w in io[:6]] == [buf + 4 * k for k in range(5, -1, -1)]  # back to front
    assert [w.mem_after for w in io[:6]] == [6, 5, 4, 3, 2, 1] and vm.read(buf, 6).tolist() == [1, 2, 3, 4, 5, 6]                                  # ... in stream order
    assert [(w.rd, w.rd_before, w.rd_after) for w in io] == [(A1, 6 - k, 5 - k) for k in range(6)] + [(A1, 0, 0)]
    assert all(w.rs2_value == buf for w in io)                       # every ecall cycle reads a7 and a0
    halt = vm.preflight(0)[-1]
    assert (halt.insn, halt.rs1_value, halt.rs2_value, halt.rd, halt.mem_kind) == (ECALL, 0, 0, 0, r0.MEM_NONE)
    # a count in the millions (the fuzzer's input: 33 M words) costs cycles, not memory up front: the session limit ends it
    vm = r0.Vm()
    vm.load(0x1000, flat(LI(2, 0x02001000), ADDI(A1, 2, 4), ADDI(A7, 0, 1), ADDI(A0, 2, 0), ECALL))
    vm.set_pc(0x1000)
    assert vm.run(segment_po2=10, page_in_cycles=16, page_out_cycles=16, max_cycles=5000)[0] == r0.Vm.LIMIT and vm.cycles == 5000
    # a count no transfer may have is refused before anything moves
    vm = r0.Vm()
    vm.load(0x1000, flat(LI(A1, 0x7FFFFFF0), LI(A0, 0x2000), ADDI(A7, 0, 1), ECALL))
    vm.set_pc(0x1000)
    with pytest.raises(r0.R0HipError, match="count too large"):
        vm.run()
    # the journal window: a COMMIT from anywhere else, of a word twice, or one that leaves a hole is a guest trap -- a verifier who
    # knows only the journal's bytes must know which (address, word) pairs the COMMIT rows named
    jb = r0.JOURNAL_BASE
    halt = [ADDI(A0, 0, 0), ADDI(A7, 0, 0), ECALL]
    for body, why in (([LI(A0, 0x3000), ADDI(A1, 0, 1), ADDI(A7, 0, 2), ECALL], "outside the journal window"),
                      ([LI(A0, jb), ADDI(A1, 0, 2), ADDI(A7, 0, 2), ECALL, LI(A0, jb + 4), ADDI(A1, 0, 1), ADDI(A7, 0, 2), ECALL], "committed twice"),
                      ([LI(A0, jb + 8), ADDI(A1, 0, 1), ADDI(A7, 0, 2), ECALL], "hole in the journal")):
        vm = r0.Vm()
        vm.load(0x1000, flat(*body, *halt))
        vm.set_pc(0x1000)
        with pytest.raises(r0.R0HipError, match=why):
            vm.run()
    vm = r0.Vm()  # two commits in any order that together cover [0, 3): fine
    vm.load(0x1000, flat(LI(A0, jb), ADDI(A1, 0, 3), ADDI(A7, 0, 1), ECALL, LI(A0, jb + 4), ADDI(A1, 0, 2), ADDI(A7, 0, 2), ECALL, LI(A0, jb), ADDI(A1, 0, 1), ADDI(A7, 0, 2), ECALL, *halt))
    vm.set_pc(0x1000)
    vm.set_input([5, 6, 7])
    assert vm.run() == (0, 0) and vm.journal == struct.pack("<III", 5, 6, 7)


def test_guest_memory_is_the_low_gibibyte():
    """Addresses at or above 2^30 trap -- loads, stores, jumps, ecall buffers: the trace circuit carries a pc as one field element and
    keeps the registers right above 2^28 memory words (R0H_REG_BASE)."""
    assert r0.REG_BASE == 1 << 28
    top = 0x40000000
    for prog, why in [(flat(LI(T0, top), I(0, T0, 2, T1, 0x03)), "load outside"), (flat(LI(T0, top - 4), S(8, T1, T0, 2)), "store outside"),
                      (flat(LI(T0, top), I(0, T0, 0, 0, 0x67)), "pc outside"),
                      (flat(LI(A0, top - 8), ADDI(A1, 0, 4), ADDI(A7, 0, 1), ECALL), "ecall buffer outside"),
                      (flat(LI(A0, top), ADDI(A1, 0, 0), ADDI(A7, 0, 2), ECALL), "ecall buffer outside")]:
        vm = r0.Vm()
        vm.load(0x1000, prog)
        vm.set_pc(0x1000)
        with pytest.raises(r0.R0HipError, match=why):
            vm.run()
    vm = r0.Vm()  # the last word below the line is memory like any other
    vm.load(0x1000, flat(LI(T0, top - 4), ADDI(T1, 0, 77), S(0, T1, T0, 2), I(0, T0, 2, T2, 0x03), ADDI(A0, T2, 0), ADDI(A7, 0, 0), ECALL))
    vm.set_pc(0x1000)
    assert vm.run() == (0, 77)


def test_elf_loader():
    prog = flat(ADDI(A0, 0, 42), ADDI(A7, 0, 0), ECALL)
    code = struct.pack("<%dI" % len(prog), *prog)
    entry, vaddr = 0x10000, 0x10000
    ehdr = b"\x7fELF" + bytes([1, 1, 1, 0]) + bytes(8) + struct.pack("<HHIIIIIHHHHHH", 2, 243, 1, entry, 52, 0, 0, 52, 32, 1, 0, 0, 0)
    phdr = struct.pack("<IIIIIIII", 1, 84, vaddr, vaddr, len(code), len(code) + 64, 5, 4)
    vm = r0.Vm()
    vm.load_elf(ehdr + phdr + code)
    assert vm.pc == entry and vm.run() == (0, 42) and vm.read(vaddr + len(code), 4).tolist() == [0, 0, 0, 0]
    with pytest.raises(r0.R0HipError, match="already run"):
        vm.run()
    # .bss is not materialised (a header may claim 369 MB: found by tools/fuzz), but it does clear what an earlier segment loaded there
    big = struct.pack("<IIIIIIII", 1, 84 + 32, vaddr, vaddr, len(code), 0x1600004C, 5, 4)
    two = ehdr[:44] + struct.pack("<H", 2) + ehdr[46:]
    later = struct.pack("<IIIIIIII", 1, 84 + 32, vaddr + 0x2000, vaddr + 0x2000, len(code), len(code), 5, 4)
    vm = r0.Vm()
    vm.load_elf(two + later + big + code)   # first the small segment at +0x2000, then one whose .bss covers it
    assert vm.read(vaddr, 3).tolist() == list(prog) and vm.read(vaddr + 0x2000, 3).tolist() == [0, 0, 0] and vm.run() == (0, 42)
    for bad, why in [(b"\x7fELG" + bytes(60), "magic"), (ehdr[:18] + struct.pack("<H", 62) + ehdr[20:] + phdr + code, "RISC-V"), (ehdr + phdr[:16] + struct.pack("<I", 9999) + phdr[20:] + code, "malformed")]:
        with pytest.raises(r0.R0HipError, match=why):
            r0.Vm().load_elf(bad)
