"""The trace circuit (tools/gen_circuit.py trace, circuits/trace.r0c): a circuit whose DATA group IS the executor's preflight trace
-- what the prover commits to comes from an execution, not from a synthetic column program.
It constrains that the cycles form one contiguous run from the public first pc to the public last pc in the public number of
cycles, WHAT EVERY INSTRUCTION DOES (decode, ALU / shifter / multiplier results, branch decisions, jump targets, load / store
addresses and the narrow accesses' byte lanes, quotients and remainders, which registers and which word an ecall reads and writes) and
MEMORY CONSISTENCY over registers and memory as one address space (offline memory checking: a grand product over
r0h_prefix_products in ACCUM, timestamps ordered through radix-4 digits in DATA): what is read from a register or a word -- an
instruction word included -- is what was last written there.  It is this library's circuit for this library's executor, not
risc0's rv32im circuit (whose tap table and constraint polynomial cannot be reproduced here).  The non-gpu tests prove with the
oracle; both verifiers check.  The columns that come straight from the compact rows are restated here in numpy, independently of
csrc/trace.hpp; the derived ones (operand bits, digits, carries) are checked by evaluating every constraint on the witness
(tools/gen_circuit.py check_trace_rows names the constraint a witness breaks)."""
import os
import struct
import sys

import numpy as np
import pytest

import hyperfridge_r0_amd as r0
from conftest import circuit_path
from test_rv32im import _guest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
from gen_circuit import OPCODES, TRACE_COLUMNS, check_trace_rows  # noqa: E402

COL = {name: i for i, name in enumerate(TRACE_COLUMNS)}
P = 2013265921
REG = r0.REG_BASE
F = dict(cycle=0, pc=1, insn=2, next_pc=3, rs1=4, rs2=5, rd=6, rd_before=7, rd_after=8, mem_kind=9, mem_addr=10, mem_before=11, mem_after=12, prev=13)


def _run(n_loop=60, po2=20):
    prog, base = _guest(n_loop), 0x400
    vm = r0.Vm()
    vm.load(base, prog)
    vm.set_pc(base)
    vm.set_input([7, 0x01020304])
    assert vm.run(segment_po2=po2, keep_trace=True, boundary_rows=True) == (0, 0)
    return vm, base


PRIMARY = (["live", "bnd", "cycle", "pc", "next_pc", "insn_lo", "insn_hi"] + ["bit%d" % k for k in range(32)] + ["opc_" + n for n, _ in OPCODES]
           + ["f3_%d" % k for k in range(8)] + ["z1", "inv1", "act0", "addr0", "rs1_lo", "rs1_hi", "p0", "tw0", "z2", "inv2", "act1", "addr1", "rs2_lo", "rs2_hi", "p1", "tw1"]
           + ["zrd", "inv_rd", "act2", "addr2", "old_lo", "old_hi", "new_lo", "new_hi", "p2", "tw2", "mem_kind", "addr3", "before_lo", "before_hi", "after_lo", "after_hi", "p3", "tw3"]
           + ["addr4", "p4", "tw4"] + ["d%d_%d" % (k, i) for k in (0, 1, 2, 4) for i in range(12)])


def expand(rows, bounds, po2):
    """The columns of the DATA group that come straight from the compact rows (PRIMARY), as canonical integers, [column, row]: the
    specification of include/r0hip.h (r0h_preflight_row, r0h_preflight_bound, the trace-circuit paragraph) written out with numpy.
    (Access 3's digits are left out: rows that multiply keep carries there.)"""
    n, nr, nb = 1 << po2, len(rows), len(bounds)
    m = np.zeros((len(TRACE_COLUMNS), n), dtype=np.int64)
    inv = lambda v: pow(int(v) % P, P - 2, P)
    r = rows.astype(np.int64)
    L = slice(0, nr)
    insn, cyc = r[:, F["insn"]], r[:, F["cycle"]]
    m[COL["live"], L] = 1
    m[COL["cycle"], L] = cyc
    m[COL["pc"], L] = r[:, F["pc"]]
    m[COL["next_pc"], L] = r[:, F["next_pc"]]
    m[COL["insn_lo"], L] = insn & 0xFFFF
    m[COL["insn_hi"], L] = insn >> 16
    for k in range(32):
        m[COL["bit%d" % k], L] = (insn >> k) & 1
    for name, code in OPCODES:
        m[COL["opc_" + name], L] = (insn & 0x7F) == code
    m[COL["f3_0"]] = 1
    for k in range(8):
        m[COL["f3_%d" % k], L] = ((insn >> 12) & 7) == k
    small = np.array([0] + [inv(i) for i in range(1, 32)], dtype=np.int64)
    for k, (z, iv, act, addr, lo, hi, p, tw, shift, val) in enumerate((("z1", "inv1", "act0", "addr0", "rs1_lo", "rs1_hi", "p0", "tw0", 15, "rs1"),
                                                                      ("z2", "inv2", "act1", "addr1", "rs2_lo", "rs2_hi", "p1", "tw1", 20, "rs2"))):
        idx = (insn >> shift) & 31
        m[COL[z]] = 1
        m[COL[z], L] = idx == 0
        m[COL[iv], L] = small[idx]
        idx = np.where(insn == 0x73, (17, 10)[k], idx)  # an ecall reads a7 and a0 where its word names x0 twice
        on = idx != 0
        m[COL[act], L] = on
        m[COL[lo], L] = r[:, F[val]] & 0xFFFF
        m[COL[hi], L] = r[:, F[val]] >> 16
        m[COL[addr], L] = np.where(on, REG + idx, 0)
        m[COL[p], L] = np.where(on, r[:, F["prev"] + k], 0)
        m[COL[tw], L] = np.where(on, 5 * cyc + k + 1, 0)
    m[COL["zrd"]] = 1
    m[COL["zrd"], L] = ((insn >> 7) & 31) == 0
    m[COL["inv_rd"], L] = small[(insn >> 7) & 31]
    wr = r[:, F["rd"]] != 0
    m[COL["act2"], L] = wr
    m[COL["addr2"], L] = np.where(wr, REG + r[:, F["rd"]], 0)
    for name, f in (("old", "rd_before"), ("new", "rd_after")):
        m[COL[name + "_lo"], L] = np.where(wr, r[:, F[f]] & 0xFFFF, 0)
        m[COL[name + "_hi"], L] = np.where(wr, r[:, F[f]] >> 16, 0)
    m[COL["p2"], L] = np.where(wr, r[:, F["prev"] + 2], 0)
    m[COL["tw2"], L] = np.where(wr, 5 * cyc + 3, 0)
    mem = r[:, F["mem_kind"]] != 0
    m[COL["mem_kind"], L] = r[:, F["mem_kind"]]
    m[COL["addr3"], L] = np.where(mem, r[:, F["mem_addr"]] >> 2, 0)
    for name, f in (("before", "mem_before"), ("after", "mem_after")):
        m[COL[name + "_lo"], L] = np.where(mem, r[:, F[f]] & 0xFFFF, 0)
        m[COL[name + "_hi"], L] = np.where(mem, r[:, F[f]] >> 16, 0)
    m[COL["p3"], L] = np.where(mem, r[:, F["prev"] + 3], 0)
    m[COL["tw3"], L] = np.where(mem, 5 * cyc + 4, 0)
    m[COL["addr4"], L] = r[:, F["pc"]] >> 2
    m[COL["p4"], L] = r[:, F["prev"] + 4]
    m[COL["tw4"], L] = 5 * cyc + 5
    for k in (0, 1, 2, 4):
        diff = np.where(m[COL["tw%d" % k], L] > 0, m[COL["tw%d" % k], L] - m[COL["p%d" % k], L] - 1, 0)
        assert (diff >= 0).all() and (diff < 1 << 24).all()
        for i in range(12):
            m[COL["d%d_%d" % (k, i)], L] = (diff >> (2 * i)) & 3
    if nb:
        bb = bounds.astype(np.int64)
        B = slice(nr, nr + nb)
        m[COL["bnd"], B] = 1
        m[COL["addr3"], B] = bb[:, 0]
        m[COL["after_lo"], B], m[COL["after_hi"], B] = bb[:, 1] & 0xFFFF, bb[:, 1] >> 16     # written: the value found, timestamp 0
        m[COL["before_lo"], B], m[COL["before_hi"], B] = bb[:, 2] & 0xFFFF, bb[:, 2] >> 16   # read: the value left, at its last timestamp
        m[COL["p3"], B] = bb[:, 3]
        assert (bb[:, 0] < (1 << 28) + 32).all()
        top = bb[:, 0] >> 28                                                                 # the address: fourteen digits and the register bit
        low = bb[:, 0] - (top << 28)
        for i in range(14):
            m[COL["d%d_%d" % (i // 12, i % 12)], B] = (low >> (2 * i)) & 3
        m[COL["d1_2"], B] = top
        gap = np.concatenate([[0], bb[1:, 0] - bb[:-1, 0] - 1])
        assert (gap >= 0).all() and (gap < 1 << 30).all()
        for i in range(3):
            m[COL["d2_%d" % i], B] = (gap >> (2 * (12 + i))) & 3
    return m


def montgomery(m):
    return ((m.astype(object) << 32) % P).astype(np.uint32)


R_INV = pow(1 << 32, P - 2, P)


def canonical(words, po2):
    """witness words (Montgomery form, column-major) -> [column, row] canonical integers"""
    return (words.reshape(len(TRACE_COLUMNS), 1 << po2).astype(np.int64) * R_INV) % P


def broken(vm, k, po2, edits=()):
    """names of the constraints the witness of segment k breaks, after `edits` [(column, row, canonical value)]"""
    data, glob = vm.trace_witness(k, po2)
    m = canonical(data, po2)
    for c, r, v in edits:
        m[COL[c], r] = v % P
    return [name for name, _ in check_trace_rows(m, [int(g) * R_INV % P for g in glob])]


def test_column_list_is_the_one_the_library_fills():
    assert r0.trace_column_names() == TRACE_COLUMNS and len(TRACE_COLUMNS) == r0.TRACE_COLUMNS


def test_the_witness_is_the_preflight_trace(orc):
    vm, base = _run()
    rows, bounds = vm.preflight_arrays(0)
    n, po2 = len(rows), 10
    assert 256 < n and n + len(bounds) <= 1 << po2 and len(bounds) == vm.segments()[0].boundary_rows
    data, glob = vm.trace_witness(0, po2)
    want = montgomery(expand(rows, bounds, po2))
    got = data.reshape(r0.TRACE_COLUMNS, 1 << po2)
    primary = [COL[c] for c in PRIMARY]
    assert np.array_equal(got[primary], want[primary]), [TRACE_COLUMNS[c] for c in primary if (got[c] != want[c]).any()]
    assert broken(vm, 0, po2) == []  # ... and the derived columns satisfy every constraint
    assert [orc.dec(int(g)) for g in glob] == [0] * 8 + [base, int(rows[-1, F["next_pc"]]), n, 1, 1, 0, 0]  # ... ends in HALT(0)
    # the rows themselves: timestamps name the previous access, boundary rows are each address once, in order, with what was found and left
    last, value = {}, {}
    for w in vm.preflight(0):
        i1, i2 = ((w.insn >> 15) & 31, (w.insn >> 20) & 31) if w.insn != 0x73 else (17, 10)  # an ecall reads a7 and a0
        acc = [(REG + i1, w.rs1_value, w.rs1_value) if i1 else None, (REG + i2, w.rs2_value, w.rs2_value) if i2 else None,
               (REG + w.rd, w.rd_before, w.rd_after) if w.rd else None, (w.mem_addr >> 2, w.mem_before, w.mem_after) if w.mem_kind else None,
               (w.pc >> 2, w.insn, w.insn)]
        for k, a in enumerate(acc):
            if a is None:
                continue
            addr, before, after = a
            assert w.prev[k] == last.get(addr, 0), (w.cycle, k)
            if addr in value:
                assert value[addr][1] == before, (w.cycle, k, hex(addr))  # what is read is what was last written
            else:
                value[addr] = [before, before]
            value[addr][1] = after
            last[addr] = 5 * w.cycle + k + 1
    bl = vm.boundary(0)
    assert [b.addr for b in bl] == sorted(value) and all((b.first_value, b.last_value, b.last_ts) == (value[b.addr][0], value[b.addr][1], last[b.addr]) for b in bl)
    with pytest.raises(r0.R0HipError, match="do not fit"):
        vm.trace_witness(0, 8)


def _prover(orc, po2):
    blob = np.fromfile(circuit_path("trace"), dtype=np.uint32)
    c = orc.circuit(blob)
    code, _, _ = c.witgen(po2, 0)  # the fixed CODE columns (first / last row, row index): the program's control root comes from them
    root = c.code_root(code, po2)

    def rejected(d, g):
        s = c.prove(po2, code, d, g)
        got = c.verify(s, code_root=root)
        assert got == r0.verify_seal(blob, s, code_root=root)[:2]
        return got[0] == 4  # the constraint identity at z fails
    return blob, c, code, root, rejected


def test_an_execution_proves_and_an_altered_one_does_not(orc):
    vm, base = _run()
    rows = vm.preflight(0)
    n, po2 = len(rows), 10
    N = 1 << po2
    data, glob = vm.trace_witness(0, po2)
    blob, c, code, root, rejected = _prover(orc, po2)
    seal = c.prove(po2, code, data, glob)
    assert c.verify(seal, code_root=root) == (0, "ok")
    assert r0.verify_seal(blob, seal, code_root=root)[:2] == (0, "ok")
    enc = orc.enc

    def edit(*changes):
        d = data.copy()
        for col, row, value in changes:
            d[COL[col] * N + row] = enc(value % P)
        return d

    mid = n // 2
    assert rejected(edit(("pc", mid, 0x5000)), glob)                   # a row that starts somewhere its predecessor did not go
    assert rejected(edit(("next_pc", mid, 0x5000)), glob)              # ... or goes somewhere the next one does not start
    assert rejected(edit(("cycle", mid, mid + 1)), glob)               # a skipped cycle
    assert rejected(edit(("live", mid, 0)), glob)                      # a hole in the run
    assert rejected(edit(("live", N - 1, 1)), glob)                    # a row smuggled in after the end
    assert rejected(edit(("mem_kind", mid, 3)), glob)                  # not none / read / write
    rd = next(r for r, w in enumerate(rows) if w.mem_kind == r0.MEM_READ)
    assert rejected(edit(("after_lo", rd, (rows[rd].mem_after & 0xffff) ^ 1)), glob)  # a read that changes the word
    for k, wrong in ((8, base + 4), (9, 0x5000), (10, n - 1)):         # public inputs that do not describe this run
        g = glob.copy()
        g[k] = enc(wrong)
        assert rejected(data, g)
    g = glob.copy()
    g[0] = enc(1)                                                      # the claim words are bound by the transcript, not by a constraint:
    s = c.prove(po2, code, data, g)                                    # another claim, another (valid) seal ...
    assert c.verify(s, code_root=root) == (0, "ok") and not np.array_equal(s[:11], seal[:11])
    s[0] = seal[0]                                                     # ... which does not pass for this one
    assert c.verify(s, code_root=root)[0] != 0
    # control flow follows the instruction words
    br = next(r for r, w in enumerate(rows) if (w.insn & 0x7f) == 0x63 and w.next_pc != w.pc + 4)   # a taken branch
    assert rejected(edit(("opc_branch", br, 0)), glob)                 # ... cannot pass as an ordinary instruction
    assert rejected(edit(("bit0", br, 0)), glob)                       # ... nor can its word be changed under it (halves and opcode pin the bits)
    assert rejected(edit(("next_pc", br, rows[br].pc + 8), ("pc", br + 1, rows[br].pc + 8), ("addr4", br + 1, (rows[br].pc + 8) >> 2)), glob)  # nor go elsewhere
    alu = next(r for r, w in enumerate(rows) if (w.insn & 0x7f) == 0x13 and r > 4)
    assert rejected(edit(("next_pc", alu, rows[alu].pc + 8), ("pc", alu + 1, rows[alu].pc + 8)), glob)  # an ALU instruction that skips the next one
    assert rejected(edit(("opc_jal", alu, 1)), glob)                   # ... and cannot be flagged as a jump to get away with it
    io = next(r for r, w in enumerate(rows) if w.insn == 0x73 and w.next_pc == w.pc)  # an I/O ecall repeating: to pc or pc + 4, nowhere else
    assert rejected(edit(("next_pc", io, rows[io].pc + 8), ("pc", io + 1, rows[io].pc + 8)), glob)

    # ---- memory consistency: registers, memory and instruction words
    def next_read(reg, after):
        return next(r for r in range(after + 1, n) if ((rows[r].insn >> 15) & 31) == reg or ((rows[r].insn >> 20) & 31) == reg)

    w_row = next(r for r, w in enumerate(rows) if w.rd and r > 8 and any(((q.insn >> 15) & 31) == w.rd for q in rows[r + 1:r + 30]))
    reg = rows[w_row].rd
    r_row = next_read(reg, w_row)
    slot = "rs1" if ((rows[r_row].insn >> 15) & 31) == reg else "rs2"
    v = getattr(rows[r_row], slot + "_value")
    assert rejected(edit((slot + "_lo", r_row, (v & 0xffff) ^ 1)), glob)           # a register that changes between its write and the next read
    assert rejected(edit(("new_lo", w_row, (rows[w_row].rd_after & 0xffff) ^ 1)), glob)   # ... from either side
    assert rejected(edit(("old_hi", w_row, (rows[w_row].rd_before >> 16) ^ 1)), glob)     # a write that misstates what it overwrote
    st = next(r for r, w in enumerate(rows) if w.mem_kind == r0.MEM_WRITE and r > 20)
    assert rejected(edit(("after_lo", st, (rows[st].mem_after & 0xffff) ^ 4)), glob)      # a store whose word is not what is found there later (its boundary row)
    assert rejected(edit(("before_lo", rd, (rows[rd].mem_before & 0xffff) ^ 1), ("after_lo", rd, (rows[rd].mem_after & 0xffff) ^ 1)), glob)  # a load that sees another word
    assert rejected(edit(("insn_lo", mid, rows[mid].insn & 0xffff ^ 0x1000), ("bit12", mid, 1 - ((rows[mid].insn >> 12) & 1))), glob)  # an instruction word that is not the one in memory
    assert rejected(edit(("addr0", r_row if slot == "rs1" else next_read(reg, r_row), REG + 31)), glob)  # reading another register than the word names
    assert rejected(edit(("p4", mid, rows[mid].prev[4] + 1)), glob)                # a made-up previous timestamp
    assert rejected(edit(("tw4", mid, 5 * mid + 6)), glob)                         # ... or own timestamp
    x0 = next(r for r, w in enumerate(rows) if ((w.insn >> 15) & 31) == 0 and (w.insn & 0x7f) == 0x13)
    assert rejected(edit(("rs1_lo", x0, 5)), glob)                                 # x0 reads as zero
    assert rejected(edit(("z1", x0, 0), ("act0", x0, 1)), glob)                    # ... and cannot be declared a register access
    bounds = vm.boundary(0)
    b0 = n + 3
    assert rejected(edit(("after_lo", b0, (bounds[3].first_value & 0xffff) ^ 1)), glob)  # the first value of an address is what its first access finds
    assert rejected(edit(("p3", b0, bounds[3].last_ts + 5)), glob)
    assert rejected(edit(("addr3", b0, bounds[2].addr)), glob)                     # an address twice among the boundary rows (two histories)
    assert rejected(edit(("bnd", b0, 0)), glob)                                    # a boundary row dropped
    # a consistent lie about what an instruction computed -- the value written changed together with its result column and with every
    # later sight of it, up to the register's next write or its boundary row: memory stays consistent, the instruction does not
    lie = (rows[w_row].rd_after & 0xffff) ^ 1
    chain = [("new_lo", w_row, lie), ("res_lo", w_row, lie)]
    r = w_row
    while True:  # every later sight of that register value up to its next write
        nxt = [q for q in range(r + 1, n) if ((rows[q].insn >> 15) & 31) == reg or ((rows[q].insn >> 20) & 31) == reg or rows[q].rd == reg]
        if not nxt:
            chain.append(("before_lo", n + [b.addr for b in bounds].index(REG + reg), lie))
            break
        r = nxt[0]
        if ((rows[r].insn >> 15) & 31) == reg:
            chain.append(("rs1_lo", r, lie))
        if ((rows[r].insn >> 20) & 31) == reg:
            chain.append(("rs2_lo", r, lie))
        if rows[r].rd == reg:
            chain.append(("old_lo", r, lie))
            break
    assert rejected(edit(*chain), glob)


def forged_result(r, value, word="z"):
    """the edits of a prover that claims instruction r wrote `value`: the register's new value, the result columns and the
    range-checked word the result is read from (Z or W) with everything the always-on definitions derive from it"""
    lo, hi = value & 0xFFFF, value >> 16
    edits = [("new_lo", r, lo), ("new_hi", r, hi), ("res_lo", r, lo), ("res_hi", r, hi)]
    if word == "u":
        return edits + [("ub%d" % i, r, (value >> i) & 1) for i in range(32)]
    edits += [("%sd%d" % (word, i), r, (value >> (2 * i)) & 3) for i in range(16)]
    if word == "z":
        edits += [("ob0", r, value & 1), ("ob1", r, (value >> 1) & 1), ("eq", r, int(value == 0)), ("zinv", r, pow(lo + hi, P - 2, P) if value else 0)]
    return edits


def test_what_an_instruction_computes_is_constrained_kind_by_kind(orc):
    """Random programs over every RV32IM instruction kind (tools/soak_trace.py): the genuine witness satisfies every constraint,
    and for each kind that writes a register the most careful lie available -- another value written, the result columns and the
    range-checked word changed with it -- breaks a constraint that belongs to that instruction's unit.  Stores: another word
    written.  Branches: the other way taken."""
    from soak_trace import random_program
    rng = np.random.default_rng(21)
    seen = {}
    for trial in range(6):
        vm = r0.Vm()
        vm.load(0x1000, random_program(rng, 350))
        vm.set_pc(0x1000)
        for i in range(1, 28):
            vm.set_reg(i, int(rng.integers(0, 1 << 32)) if rng.random() < 0.7 else int(rng.choice([0, 1, 0xFFFFFFFF, 0x80000000])))
        vm.set_input([int(x) for x in rng.integers(0, 1 << 32, 8)])
        assert vm.run(segment_po2=20, keep_trace=True, boundary_rows=True, max_cycles=50_000) == (0, 0)
        seg = vm.segments()[0]
        po2 = max(9, int(np.ceil(np.log2(seg.user_cycles + seg.boundary_rows))))
        assert broken(vm, 0, po2) == []
        rows = vm.preflight(0)
        for r, w in enumerate(rows):
            op, f3, f7 = w.insn & 0x7F, (w.insn >> 12) & 7, w.insn >> 25
            kind = (op, f3, f7 if op == 0x33 else (f7 & 0x20) if (op == 0x13 and f3 == 5) else 0)
            if kind in seen or r == len(rows) - 1:
                continue
            if w.rd:
                from_w = op in (0x6F, 0x67) or (op in (0x13, 0x33) and f3 == 5 and f7 != 1) or (op == 0x33 and f7 == 1 and f3 in (1, 2, 3))
                from_u = op == 0x33 and f7 == 1 and f3 in (4, 5)  # a quotient
                bad = broken(vm, 0, po2, forged_result(r, w.rd_after ^ 0x10, "u" if from_u else "w" if from_w else "z"))
                assert bad, hex(w.insn)
                if op != 0x73:
                    assert not any(name.startswith(("rd:", "run:", "accum", "bit:", "digit:")) for name in bad), (hex(w.insn), bad)
                seen[kind] = bad
            elif op == 0x23:
                bad = broken(vm, 0, po2, [("after_lo", r, (w.mem_after & 0xFFFF) ^ 0x100)])
                assert bad and all(name.startswith(("sb:", "sh:", "sw:")) for name in bad), (hex(w.insn), bad)
                seen[kind] = bad
            elif op == 0x63:
                other = w.pc + 4 if w.next_pc != w.pc + 4 else (w.pc + 8) & 0xFFFFFFFF
                bad = broken(vm, 0, po2, [("next_pc", r, other), ("pc", r + 1, other), ("addr4", r + 1, other >> 2)])
                assert "next:branch" in bad, (hex(w.insn), bad)
                seen[kind] = bad
    ops = {k[0] for k in seen}
    assert {0x37, 0x17, 0x6F, 0x67, 0x63, 0x03, 0x23, 0x13, 0x33, 0x73} <= ops and len(seen) >= 50, sorted(seen)
    assert {(0x33, f3, 1) for f3 in range(8)} <= set(seen) and {(0x33, 0, 0x20), (0x33, 5, 0x20), (0x13, 5, 0x20), (0x13, 1, 0)} <= set(seen)
    assert {(0x03, f3, 0) for f3 in (0, 1, 2, 4, 5)} | {(0x23, f3, 0) for f3 in range(3)} | {(0x63, f3, 0) for f3 in (0, 1, 4, 5, 6, 7)} <= set(seen)


def test_an_ecall_row_does_what_its_function_says(orc):
    """READ_WORDS / COMMIT / CYCLES / HALT rows: a7 and a0 are the two registers read, the transfers count a1 down and touch the
    word a0 + 4 (a1 - 1), HALT touches nothing.  What is constrained is which registers and which word move, not the words
    themselves (input is the host's to choose, the journal is bound by the claim's output digest): the value CYCLES writes and the
    word READ_WORDS stores may be anything that is a 32-bit word."""
    from test_rv32im import ADDI, A0, A1, A7, ECALL, LI, flat
    buf = 0x3000
    prog = flat(LI(A0, buf), ADDI(A1, 0, 3), ADDI(A7, 0, 1), ECALL, ADDI(A1, 0, 2), ADDI(A7, 0, 2), ECALL, ADDI(A7, 0, 3), ECALL,
                LI(A0, buf), ADDI(A1, 0, 0), ADDI(A7, 0, 1), ECALL, ADDI(A0, 0, 5), ADDI(A7, 0, 0), ECALL)
    vm = r0.Vm()
    vm.load(0x1000, prog)
    vm.set_pc(0x1000)
    vm.set_input([11, 22, 33])
    assert vm.run(segment_po2=20, keep_trace=True, boundary_rows=True) == (0, 5) and vm.journal == struct.pack("<II", 11, 22)
    po2 = 9
    assert broken(vm, 0, po2) == []
    rows = vm.preflight(0)
    sysrows = [(r, w) for r, w in enumerate(rows) if w.insn == 0x73]
    assert [w.rs1_value for _, w in sysrows] == [1, 1, 1, 1, 2, 2, 2, 3, 1, 0]
    rd0 = sysrows[0][0]      # READ_WORDS, a1 = 3: writes word 2 of the buffer
    assert rows[rd0].mem_addr == buf + 8 and rows[rd0].mem_after == 33
    assert "ecall:addr" in broken(vm, 0, po2, [("addr3", rd0, (buf + 4) >> 2)])                                  # another word of the buffer
    assert "ecall:count_lo" in broken(vm, 0, po2, forged_result(rd0, 1))                                             # a1 skips a step
    assert "next:ecall" in broken(vm, 0, po2, [("next_pc", rd0, rows[rd0].pc + 4), ("pc", rd0 + 1, rows[rd0].pc + 4), ("addr4", rd0 + 1, (rows[rd0].pc + 4) >> 2)])
    assert "ecall:mem" in broken(vm, 0, po2, [("mem_kind", rd0, 1)])                                             # READ_WORDS does not read memory
    assert broken(vm, 0, po2, [("addr0", rd0, REG + 16)]) == ["rs1:addr"]                                        # the function is in a7, nowhere else
    assert broken(vm, 0, po2, [("addr1", rd0, REG + 12)]) == ["rs2:addr"]
    word = [("after_lo", rd0, 0x1234), ("after_hi", rd0, 0x5678)] + [("wd%d" % i, rd0, (0x56781234 >> (2 * i)) & 3) for i in range(16)]
    assert broken(vm, 0, po2, word) == []                                                                        # the word read in is the host's to choose
    assert "digit:wd3" in broken(vm, 0, po2, word + [("wd3", rd0, 4), ("after_lo", rd0, 0x1234 + 64)])           # ... as long as it is a word
    done = sysrows[3][0]     # a1 = 0: falls through, touches nothing
    assert (rows[done].mem_kind, rows[done].next_pc, rows[done].rd, rows[done].rd_after) == (0, rows[done].pc + 4, A1, 0)
    assert "ecall:mem" in broken(vm, 0, po2, [("mem_kind", done, 2)])
    cyc = sysrows[7][0]
    other = (123456 << 2) | (rows[cyc].rd_after & 3)  # (Z's two low bits also select a byte of U; kept, so nothing else has to follow)
    anyword = [(c, cyc, v) for c, v in (("new_lo", other & 0xFFFF), ("new_hi", other >> 16), ("res_lo", other & 0xFFFF), ("res_hi", other >> 16))]
    assert rows[cyc].rd == A0 and broken(vm, 0, po2, anyword + [("zd%d" % i, cyc, (other >> (2 * i)) & 3) for i in range(16)]) == []  # CYCLES: any word
    assert "ecall:rd" in broken(vm, 0, po2, [("addr2", cyc, REG + 11)])                                          # ... into a0
    halt = sysrows[-1][0]
    assert "ecall:mem" in broken(vm, 0, po2, [("mem_kind", halt, 2)])                                            # HALT does not write memory
    assert "ecall:act2" in broken(vm, 0, po2, [("act2", halt, 1), ("addr2", halt, REG + 11), ("tw2", halt, 5 * halt + 3)])  # ... nor a register
    # the public inputs say how the segment ends: HALT with exit code 5
    data, glob = vm.trace_witness(0, po2)
    g = [int(x) * R_INV % P for x in glob]
    assert g[8:] == [0x1000, rows[-1].next_pc, len(rows), 1, 1, 5, 0]
    m = canonical(data, po2)
    for k, wrong, name in ((11, 2, "exit:end_kind"), (13, 6, "exit:end_lo"), (14, 1, "exit:end_hi"), (12, 0, "exit:end_is")):
        bad = list(g)
        bad[k] = wrong
        assert name in [nm for nm, _ in check_trace_rows(m, bad)], name
    cut = list(g)
    cut[11:15] = [0, 0, 0, 0]                                                                                   # "this segment was merely cut"
    assert "exit:end_is" in [nm for nm, _ in check_trace_rows(m, cut)]
    # ... and a HALT is the last cycle of its segment: nothing runs after it
    assert "exit:last_cycle" in broken(vm, 0, po2, [("live", len(rows), 1)])
    # an unknown function number has no satisfying row (the executor traps on it)
    assert "ecall:fn_max" in broken(vm, 0, po2, [("rs1_lo", halt, 5), ("ub0", halt, 1), ("ub2", halt, 1)])


def test_division_in_all_its_corners(orc):
    """DIV / DIVU / REM / REMU on operands of every sign, by zero, -2^31 / -1: the genuine rows satisfy the constraints; the other
    quotient-remainder pair that also satisfies dividend = quotient x divisor + remainder (quotient + 1, remainder - divisor) does not."""
    from test_rv32im import ADDI, A0, A7, ECALL, LI, R, flat
    pairs = [(7, 2), (-7, 2), (7, -2), (-7, -2), (0, 5), (5, 0), (-5, 0), (0, 0), (-2**31, -1), (-2**31, 1), (2**31 - 1, -1), (1, -2**31), (-2**31, -2**31),
             (0xFFFFFFFF, 0xFFFFFFFF), (0xFFFFFFFF, 1), (123456789, 1000), (-123456789, 1000), (6, 3), (-6, 3), (0x80000000, 0x7FFFFFFF), (5, -2**31)]
    body = []
    for a_, b_ in pairs:
        body += flat(LI(5, a_ & 0xFFFFFFFF), LI(6, b_ & 0xFFFFFFFF), [R(1, 6, 5, f3, 7 + f3) for f3 in (4, 5, 6, 7)])
    vm = r0.Vm()
    vm.load(0x1000, flat(body, ADDI(A0, 0, 0), ADDI(A7, 0, 0), ECALL))
    vm.set_pc(0x1000)
    assert vm.run(segment_po2=20, keep_trace=True, boundary_rows=True) == (0, 0)
    po2 = 10
    assert broken(vm, 0, po2) == []
    rows = vm.preflight(0)
    data, glob = vm.trace_witness(0, po2)
    m0, g = canonical(data, po2), [int(x) * R_INV % P for x in glob]
    M32 = 0xFFFFFFFF
    checked = 0
    for r, w in enumerate(rows):
        if (w.insn & 0x7F) != 0x33 or (w.insn >> 25) != 1 or ((w.insn >> 12) & 7) < 4:
            continue
        f3, a_, b_ = (w.insn >> 12) & 7, w.rs1_value, w.rs2_value
        signed = f3 in (4, 6)
        sx = lambda v: v - (1 << 32) if signed and v >> 31 else v
        if b_ == 0:
            q, rem = M32, a_
        elif signed and a_ == 0x80000000 and b_ == M32:
            q, rem = a_, 0
        else:
            qq = abs(sx(a_)) // abs(sx(b_)) * (1 if (sx(a_) < 0) == (sx(b_) < 0) else -1)  # truncating
            q, rem = qq & M32, (sx(a_) - qq * sx(b_)) & M32
        assert w.rd_after == (q if f3 in (4, 5) else rem), (hex(w.insn), a_, b_)
        assert [int(m0[COL["ub%d" % i], r]) for i in range(32)] == [(q >> i) & 1 for i in range(32)]        # U: the quotient
        assert sum(int(m0[COL["zd%d" % i], r]) << (2 * i) for i in range(16)) == rem                           # Z: the remainder
        if b_ == 0 or (signed and a_ == 0x80000000 and b_ == M32):
            continue
        # the neighbouring solution of the division identity: quotient + 1, remainder - divisor (the row's other columns re-derived by
        # hand would take a second witness generator; here the identity's own columns are edited and the comparison must object)
        q2, rem2 = (q + 1) & M32, (rem - b_) & M32
        m = m0.copy()
        for i in range(32):
            m[COL["ub%d" % i], r] = (q2 >> i) & 1
        for i in range(16):
            m[COL["zd%d" % i], r] = (rem2 >> (2 * i)) & 3
        m[COL["ob0"], r], m[COL["ob1"], r] = rem2 & 1, (rem2 >> 1) & 1
        m[COL["c1"], r], m[COL["c0"], r] = rem2 >> 31, (rem2 >> 30) & 1
        lo, hi = (q2 if f3 in (4, 5) else rem2) & 0xFFFF, (q2 if f3 in (4, 5) else rem2) >> 16
        for c, val in (("res_lo", lo), ("res_hi", hi), ("new_lo", lo), ("new_hi", hi)):
            m[COL[c], r] = val
        bad = [name for name, where in check_trace_rows(m, g) if r in where]
        assert any(name.startswith("div:") for name in bad), (a_, b_, f3, bad)
        checked += 1
    assert checked >= 60


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
        vm.run(segment_po2=20, keep_trace=True, boundary_rows=True, max_cycles=200)
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


def test_every_segment_of_a_cut_run_proves_and_the_boundary_values_chain(orc):
    """A run cut into several segments: each proves on its own; what a segment leaves in an address is what the next one that
    touches it finds (the boundary rows' first / last values) -- the link the circuit itself does not make (include/r0hip.h)."""
    vm, base = _run(700, po2=10)
    segs = vm.segments()
    assert len(segs) >= 4
    blob, c, code, root, _ = _prover(orc, 10)
    left = {}
    for k, s in enumerate(segs):
        assert s.user_cycles + s.boundary_rows <= 1 << 10 and s.boundary_rows == len(vm.boundary(k))
        data, glob = vm.trace_witness(k, 10, claim_globals=vm.claims()[k].globals())
        seal = c.prove(10, code, data, glob)
        assert c.verify(seal, code_root=root) == (0, "ok"), k
        last = k == len(segs) - 1
        assert [orc.dec(int(g)) for g in glob[8:]] == [s.pre.pc, s.post.pc, s.user_cycles] + ([1, 1, 0, 0] if last else [0, 0, 0, 0])  # HALT(0) ends the last one
        for b in vm.boundary(k):
            if b.addr in left:
                assert left[b.addr] == b.first_value, (k, hex(b.addr))
            left[b.addr] = b.last_value
    assert sum(s.user_cycles for s in segs) == vm.cycles


@pytest.mark.gpu
@pytest.mark.parametrize("n_loop,po2", [(60, 10), (9000, 17)])
def test_the_device_expands_and_proves_an_execution_trace_word_for_word_like_the_cpu_side(hal, orc, n_loop, po2):
    """The same on the GPU: the compact rows are uploaded and expanded by the device kernel (r0h_trace_witgen) -- the DATA group
    equals the host reference's word for word (and the numpy restatement's in the columns that has); CODE columns generated on the device; the seal equal to
    the CPU port's and accepted by both verifiers bound to the control root; a row that breaks the run is rejected."""
    vm, base = _run(n_loop)
    rows, bounds = vm.preflight_arrays(0)
    n = len(rows)
    assert (1 << (po2 - 2)) < n + len(bounds) <= (1 << po2)
    data, glob = vm.trace_witness(0, po2)
    blob = np.fromfile(circuit_path("trace"), dtype=np.uint32)
    c = orc.circuit(blob)
    gc = hal.load_circuit(blob)  # eval_check compiled in-process (hipRTC)
    code, synthetic, _ = hal.witgen(gc, po2, 0)
    synthetic.free()
    dev, dglob = hal.trace_witgen(rows, bounds, po2)
    assert np.array_equal(dglob, glob)
    got = dev.to_host()
    assert np.array_equal(got, data)
    if po2 <= 12:
        primary = [COL[c] for c in PRIMARY]
        assert np.array_equal(got.reshape(r0.TRACE_COLUMNS, -1)[primary], montgomery(expand(rows, bounds, po2))[primary])
    cc = hal.code_commit(gc, po2, code)
    seal = hal.prove_segment(gc, po2, cc, dev, glob)
    root = cc.root()
    assert c.verify(seal, code_root=root) == (0, "ok") and r0.verify_seal(blob, seal, code_root=root)[:2] == (0, "ok")
    ocode, _, _ = c.witgen(po2, 0)
    assert np.array_equal(ocode, code.to_host())
    assert np.array_equal(seal, c.prove(po2, ocode, data, glob))
    # a register value that changes between a write and the next read: rejected by both verifiers
    bad = rows.copy()
    k = next(r for r in range(20, n) if (bad[r, F["insn"]] >> 15) & 31)
    bad[k, F["rs1"]] ^= 1
    dev2, _ = hal.trace_witgen(bad, bounds, po2)
    seal = hal.prove_segment(gc, po2, cc, dev2, glob)
    assert c.verify(seal, code_root=root)[0] == 4 and r0.verify_seal(blob, seal, code_root=root)[0] == 4
    bad = rows.copy()
    bad[n // 2, F["pc"]] = 0x5000
    dev2.free()
    dev2, _ = hal.trace_witgen(bad, bounds, po2)
    seal = hal.prove_segment(gc, po2, cc, dev2, glob)
    assert c.verify(seal, code_root=root)[0] == 4 and r0.verify_seal(blob, seal, code_root=root)[0] == 4
    # a wrong result in a register nobody looks at before it is overwritten: memory stays consistent, the instruction does not
    # (a random program: the loop above reads everything it writes)
    from soak_trace import dead_write_lie, random_program
    rng = np.random.default_rng(po2)
    vm2 = r0.Vm()
    vm2.load(0x1000, random_program(rng, 150))
    vm2.set_pc(0x1000)
    for i in range(1, 28):
        vm2.set_reg(i, int(rng.integers(0, 1 << 32)))
    vm2.set_input([1, 2, 3, 4, 5, 6, 7, 8])
    assert vm2.run(segment_po2=20, keep_trace=True, boundary_rows=True, max_cycles=50_000) == (0, 0)
    rows2, bounds2 = vm2.preflight_arrays(0)
    assert len(rows2) + len(bounds2) <= 1 << po2
    dev2.free()
    dev2, glob2 = hal.trace_witgen(rows2, bounds2, po2)
    assert c.verify(hal.prove_segment(gc, po2, cc, dev2, glob2), code_root=root) == (0, "ok")
    lie = dead_write_lie(rows2, bounds2, rng)
    assert lie is not None
    hal.trace_witgen(lie[0], lie[1], po2, into=dev2)
    seal = hal.prove_segment(gc, po2, cc, dev2, glob2)
    assert c.verify(seal, code_root=root)[0] == 4 and r0.verify_seal(blob, seal, code_root=root)[0] == 4
    with pytest.raises(r0.R0HipError, match="do not fit"):
        hal.trace_witgen(rows, bounds, 9 if po2 > 10 else 8)
    cc.free(); code.free(); dev.free(); dev2.free(); gc.free()
