"""The trace circuit (tools/trace_circuit.py, circuits/trace.r0c, version 5): a circuit whose DATA group IS the executor's preflight
trace -- what the prover commits to comes from an execution, not from a synthetic column program.
It constrains that the cycles form one contiguous run from the public first pc to the public last pc in the public number of
cycles, WHAT EVERY INSTRUCTION DOES (decode, ALU / shifter / multiplier results, branch decisions, jump targets, load / store
addresses and the narrow accesses' byte lanes, quotients and remainders, which registers and which word an ecall reads and writes),
MEMORY CONSISTENCY over registers and memory as one address space, and -- round 4 -- what ties the segments of a session to one
another, to the program image and to the journal (the session-wide memory argument).  Range checks, byte logic, memory tuples and
session tuples are all fractions of a log-derivative argument (lookups into two 2^16-row tables of the CODE group; running sums in
ACCUM).  It is this library's circuit for this library's executor, not risc0's rv32im circuit (whose tap table and constraint
polynomial cannot be reproduced here).  The non-gpu tests prove with the oracle; both verifiers check.  The columns that come
straight from the compact rows are restated here in numpy, independently of csrc/trace.hpp; the derived ones are checked by
evaluating every constraint and every fraction on the witness (tools/gen_circuit.py check_trace_rows names what a witness breaks)."""
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
import trace_circuit as tc  # noqa: E402
import session_by_hand as sbh  # noqa: E402

COL = {name: i for i, name in enumerate(TRACE_COLUMNS)}
P = 2013265921
REG = r0.REG_BASE
PO2 = r0.TRACE_MIN_PO2
F = dict(cycle=0, pc=1, insn=2, next_pc=3, rs1=4, rs2=5, rd=6, rd_before=7, rd_after=8, mem_kind=9, mem_addr=10, mem_before=11, mem_after=12, prev=13)
STAMP = {0: 2, 1: 3, 2: 4, 3: 5, 4: 1}  # access (x[rs1], x[rs2], x[rd], memory, fetch) -> its place in the cycle


def _run(n_loop=60, po2=20):
    prog, base = _guest(n_loop), 0x400
    vm = r0.Vm()
    vm.load(base, prog)
    vm.set_pc(base)
    vm.set_input([7, 0x01020304])
    assert vm.run(segment_po2=po2, keep_trace=True, boundary_rows=True) == (0, 0)
    return vm, base


PRIMARY = (["live", "bnd", "cycle", "pc", "next_pc"] + ["opc_" + n for n, _ in OPCODES[:-1]] + ["f3_%d" % k for k in range(1, 8)]
           + ["rd0", "rdA", "rdB", "r10", "r1A", "r1B", "r20", "r2A", "r2B", "b25", "f7A", "f7B", "b30", "b31"]
           + ["rs1_lo", "rs1_hi", "dl0", "dh0", "rs2_lo", "rs2_hi", "dl1", "dh1", "zrd", "inv_rd", "act2", "old_lo", "old_hi", "dl2", "dh2"]
           + ["mem_wr", "top", "addr3", "before_lo", "before_hi", "after_lo", "after_hi", "p3", "dl4", "dh4", "fimg"])


def expand(rows, bounds, po2, number=1, closing=True):
    """The columns of the DATA group that come straight from the compact rows (PRIMARY), as canonical integers, [column, row]: the
    specification of include/r0hip.h (r0h_preflight_row, r0h_preflight_bound, the trace-circuit paragraph) written out with numpy.
    (Access 3's timestamp limbs are left out: rows that multiply keep a carry there.)"""
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
    for name, code in OPCODES[:-1]:  # (FENCE, the last of the list, has no column: it is `live` minus the others)
        m[COL["opc_" + name], L] = (insn & 0x7F) == code
    for k in range(1, 8):            # (funct3 = 0 likewise: one minus the others)
        m[COL["f3_%d" % k], L] = ((insn >> 12) & 7) == k
    for stem, shift in (("rd", 7), ("r1", 15), ("r2", 20)):
        idx = (insn >> shift) & 31
        m[COL[stem + "0"] if stem != "rd" else COL["rd0"], L] = idx & 1
        m[COL[stem + "A"], L] = (idx >> 1) & 3
        m[COL[stem + "B"], L] = idx >> 3
    m[COL["b25"], L], m[COL["f7A"], L], m[COL["f7B"], L], m[COL["b30"], L], m[COL["b31"], L] = (insn >> 25) & 1, (insn >> 26) & 3, (insn >> 28) & 3, (insn >> 30) & 1, insn >> 31
    small = np.array([0] + [inv(i) for i in range(1, 32)], dtype=np.int64)
    for k, (lo, hi, dl, dh, val) in enumerate((("rs1_lo", "rs1_hi", "dl0", "dh0", "rs1"), ("rs2_lo", "rs2_hi", "dl1", "dh1", "rs2"))):
        m[COL[lo], L] = r[:, F[val]] & 0xFFFF          # every cycle reads two registers, x0 included
        m[COL[hi], L] = r[:, F[val]] >> 16
        diff = 5 * cyc + STAMP[k] - r[:, F["prev"] + k] - 1
        assert (diff >= 0).all() and (diff < 1 << 24).all()
        m[COL[dl], L], m[COL[dh], L] = diff & 0xFFFF, diff >> 16
    m[COL["zrd"]] = 1
    m[COL["zrd"], L] = ((insn >> 7) & 31) == 0
    m[COL["inv_rd"], L] = small[(insn >> 7) & 31]
    wr = r[:, F["rd"]] != 0
    m[COL["act2"], L] = wr
    m[COL["old_lo"], L] = np.where(wr, r[:, F["rd_before"]] & 0xFFFF, 0)
    m[COL["old_hi"], L] = np.where(wr, r[:, F["rd_before"]] >> 16, 0)
    diff = np.where(wr, 5 * cyc + STAMP[2] - r[:, F["prev"] + 2] - 1, 0)
    m[COL["dl2"], L], m[COL["dh2"], L] = diff & 0xFFFF, diff >> 16
    mem = r[:, F["mem_kind"]] != 0
    m[COL["mem_wr"], L] = r[:, F["mem_kind"]] == r0.MEM_WRITE
    m[COL["addr3"], L] = np.where(mem, r[:, F["mem_addr"]] >> 2, 0)
    for name, f in (("before", "mem_before"), ("after", "mem_after")):
        m[COL[name + "_lo"], L] = np.where(mem, r[:, F[f]] & 0xFFFF, 0)
        m[COL[name + "_hi"], L] = np.where(mem, r[:, F[f]] >> 16, 0)
    m[COL["p3"], L] = np.where(mem, r[:, F["prev"] + 3], 0)
    diff = 5 * cyc + STAMP[4] - r[:, F["prev"] + 4] - 1
    m[COL["dl4"], L], m[COL["dh4"], L] = diff & 0xFFFF, diff >> 16
    if nb:
        bb = bounds.astype(np.int64)
        B = slice(nr, nr + nb)
        m[COL["bnd"], B] = 1
        m[COL["addr3"], B] = bb[:, 0]
        m[COL["after_lo"], B], m[COL["after_hi"], B] = bb[:, 1] & 0xFFFF, bb[:, 1] >> 16     # written: the value found, timestamp 0
        m[COL["before_lo"], B], m[COL["before_hi"], B] = bb[:, 2] & 0xFFFF, bb[:, 2] >> 16   # read: the value left, at its last timestamp
        m[COL["p3"], B] = bb[:, 3]
        assert (bb[:, 0] < (1 << 28) + 32).all()
        top = bb[:, 0] >> 28                                                                 # the address: two limbs and the register bit
        low = bb[:, 0] - (top << 28)
        m[COL["top"], B] = top
        m[COL["dl0"], B], m[COL["dl1"], B] = low & 0xFFFF, low >> 16
        m[COL["dh0"], B] = np.where(top == 1, 8 * (low & 0xFFFF), 0)
        m[COL["dl2"], B] = number - bb[:, 4] - 1                                             # the segment that held the address before: an earlier one
        if closing:
            m[COL["old_lo"], B], m[COL["old_hi"], B] = bb[:, 5] & 0xFFFF, bb[:, 5] >> 16     # the address's initial value
            m[COL["fimg"], B] = bb[:, 6] & 1
    return m


def montgomery(m):
    return ((m.astype(object) << 32) % P).astype(np.uint32)


R_INV = pow(1 << 32, P - 2, P)


def canonical(words, po2):
    """witness words (Montgomery form, column-major) -> [column, row] canonical integers"""
    return (words.reshape(len(TRACE_COLUMNS), 1 << po2).astype(np.int64) * R_INV) % P


def broken(vm, k, po2=PO2, edits=(), session_extra=None):
    """names of the constraints / fractions the witness of segment k breaks after `edits` [(column, row, canonical value)]; the
    forger recounts the multiplicities (a looked-up value outside its table is reported as such)"""
    data, glob = vm.trace_witness(k, po2)
    m = canonical(data, po2)
    g = [int(x) * R_INV % P for x in glob]
    for c, r, v in edits:
        m[COL[c], r] = v % P
    if edits:
        try:
            tc.multiplicities(m, g)
        except ValueError as e:
            return [str(e).split(":")[0]]
    return [name for name, _ in check_trace_rows(m, g, session_extra=session_extra)]


@pytest.fixture(scope="module")
def prover(orc):
    blob = np.fromfile(circuit_path("trace"), dtype=np.uint32)
    return blob, sbh.OracleProver(orc.circuit(blob))


def seal_of(orc, prover, po2, data, glob, seed=3):
    """one seal outside a session: any challenge will do (the late public inputs are the prover's to fill)"""
    blob, pv = prover
    rng = np.random.default_rng(seed)
    glob = glob.copy()
    glob[r0.TRACE_GAMMA:r0.TRACE_GAMMA + 16] = [orc.enc(int(v)) for v in rng.integers(0, P, 16)]
    glob = pv.totals(po2, data, glob)
    return pv.prove(po2, data, glob), glob


def test_column_list_is_the_one_the_library_fills():
    assert r0.trace_column_names() == TRACE_COLUMNS and len(TRACE_COLUMNS) == r0.TRACE_COLUMNS == 128
    assert tc.TRACE_GLOBALS == r0.TRACE_GLOBALS and tc.G_GAMMA == r0.TRACE_GAMMA and tc.G_SUM == r0.TRACE_SUM and tc.MIN_PO2 == r0.TRACE_MIN_PO2


def test_the_witness_is_the_preflight_trace(orc):
    vm, base = _run()
    rows, bounds = vm.preflight_arrays(0)
    n = len(rows)
    assert 256 < n and n + len(bounds) <= 1 << PO2 and len(bounds) == vm.segments()[0].boundary_rows and vm.segments()[0].closing == 1
    data, glob = vm.trace_witness(0, PO2)
    want = montgomery(expand(rows, bounds, PO2))
    got = data.reshape(r0.TRACE_COLUMNS, 1 << PO2)
    primary = [COL[c] for c in PRIMARY]
    assert np.array_equal(got[primary], want[primary]), [TRACE_COLUMNS[c] for c in primary if (got[c] != want[c]).any()]
    assert broken(vm, 0) == []  # ... and the derived columns satisfy every constraint, every lookup is in its table, every tuple read was written
    bl = vm.boundary(0)
    assert [orc.dec(int(g)) for g in glob[8:20]] == [base, int(rows[-1, F["next_pc"]]), n, 1, 1, 0, 0, 1, 1, 0, bl[0].addr, bl[-1].addr]  # ... ends in HALT(0); segment 1 closes its session
    assert not glob[20:].any()  # the late public inputs are the session's to fill
    # the multiplicity columns: the oracle counts the same lookups
    assert np.array_equal(orc.circuit(np.fromfile(circuit_path("trace"), dtype=np.uint32)).logup_multiplicities(PO2, data, glob), data)
    # the rows themselves: timestamps name the previous access, boundary rows are each address once, in order, with what was found and left
    last, value = {}, {}
    for w in vm.preflight(0):
        i1, i2 = ((w.insn >> 15) & 31, (w.insn >> 20) & 31) if w.insn != 0x73 else (17, 10)  # an ecall reads a7 and a0
        acc = [(REG + i1, w.rs1_value, w.rs1_value), (REG + i2, w.rs2_value, w.rs2_value),
               (REG + w.rd, w.rd_before, w.rd_after) if w.rd else None, (w.mem_addr >> 2, w.mem_before, w.mem_after) if w.mem_kind else None,
               (w.pc >> 2, w.insn, w.insn)]
        for k in (4, 0, 1, 2, 3):  # the fetch is the cycle's earliest access
            if acc[k] is None:
                continue
            addr, before, after = acc[k]
            assert w.prev[k] == last.get(addr, 0), (w.cycle, k)
            if addr in value:
                assert value[addr][1] == before, (w.cycle, k, hex(addr))  # what is read is what was last written
            else:
                value[addr] = [before, before]
            value[addr][1] = after
            last[addr] = 5 * w.cycle + STAMP[k]
    image = {(base >> 2) + i: w for i, w in enumerate(_guest(60))}
    assert [b.addr for b in bl] == sorted(set(value) | set(image))  # closing: every address touched and every image word
    for b in bl:
        if b.addr in value:
            assert (b.first_value, b.last_value, b.last_ts) == (value[b.addr][0], value[b.addr][1], last[b.addr])
        assert (b.prev_seg, b.init_value, b.flags) == (0, image.get(b.addr, 0), int(b.addr in image))
    with pytest.raises(r0.R0HipError, match="outside"):
        vm.trace_witness(0, 12)  # the lookup tables have 2^16 rows


def test_an_execution_proves_and_an_altered_one_does_not(orc, prover):
    blob, pv = prover
    vm, base = _run()
    rows = vm.preflight(0)
    n, N = len(rows), 1 << PO2
    data, glob = vm.trace_witness(0, PO2)
    root = pv.control_root(PO2)
    seal, sealed = seal_of(orc, prover, PO2, data, glob)
    assert pv.oc.verify(seal, code_root=root) == (0, "ok") and r0.verify_seal(blob, seal, code_root=root)[:2] == (0, "ok")
    assert np.array_equal(r0.control_root_host(blob, PO2), root)  # what a verifier derives from the blob alone
    enc = orc.enc

    def rejected(changes=(), g=None, recount=True):  # the full round trip: prove the altered witness, both verifiers say "constraint check mismatch at z"
        d = data.copy()
        for col, row, value in changes:
            d[COL[col] * N + row] = enc(value % P)
        if recount:
            d = pv.oc.logup_multiplicities(PO2, d, glob)
        s, _ = seal_of(orc, prover, PO2, d, glob if g is None else g)
        got = pv.oc.verify(s, code_root=root)
        assert got == r0.verify_seal(blob, s, code_root=root)[:2]
        return got[0] == 4

    mid = n // 2
    assert rejected([("pc", mid, 0x5000)])                                    # a row that starts somewhere its predecessor did not go
    rd = next(r for r, w in enumerate(rows) if w.mem_kind == r0.MEM_READ)
    assert rejected([("before_lo", rd, (rows[rd].mem_before & 0xffff) ^ 1), ("after_lo", rd, (rows[rd].mem_after & 0xffff) ^ 1)])  # a load that sees another word: the tuple it reads was never written
    assert rejected([("m16", 7, 5)], recount=False)                           # a multiplicity that is not the count: the running sum does not close
    g = glob.copy()
    g[10] = enc(n - 1)
    assert rejected(g=g)                                                      # public inputs that do not describe this run
    s2 = seal.copy()
    s2[r0.TRACE_SUM] = enc(orc.dec(int(s2[r0.TRACE_SUM])) + 1)                # the segment's session sum is bound by the transcript and by the wrap-around constraint
    assert pv.oc.verify(s2, code_root=root)[0] != 0 and r0.verify_seal(blob, s2, code_root=root)[0] != 0
    # everything else through the checker that names what breaks (a proof of such a witness fails the same way: the checker evaluates
    # the circuit's own polynomials and fractions)
    B = lambda *e: broken(vm, 0, PO2, list(e))
    assert "run:pc" in B(("next_pc", mid, 0x5000))                            # ... or goes somewhere the next one does not start
    assert "run:cycle" in B(("cycle", mid, mid + 1))                          # a skipped cycle
    assert "run:after_live" in B(("live", mid, 0))                            # a hole in the run
    assert B(("live", N - 1, 1))                                              # a row smuggled in after the end
    assert "bit:mem_wr" in B(("mem_wr", mid, 2))
    assert "mem:keeps_lo" in B(("after_lo", rd, (rows[rd].mem_after & 0xffff) ^ 1))  # a read that changes the word
    data_m = canonical(data, PO2)
    gl = [int(x) * R_INV % P for x in glob]
    for k, wrong, name in ((8, base + 4, "first:pc"), (9, 0x5000, "end:pc"), (10, n - 1, "end:cycles")):
        bad = list(gl)
        bad[k] = wrong
        assert name in [nm for nm, _ in check_trace_rows(data_m, bad)]
    # control flow follows the instruction words
    br = next(r for r, w in enumerate(rows) if (w.insn & 0x7f) == 0x63 and w.next_pc != w.pc + 4)   # a taken branch
    assert B(("opc_branch", br, 0))                                           # ... cannot pass as an ordinary instruction
    assert any(x.startswith("sum:fetch") for x in B(("rd0", br, 1 - ((rows[br].insn >> 7) & 1))))   # ... nor can its word be changed under it: the fetch reads memory
    assert "next:branch" in B(("next_pc", br, rows[br].pc + 8), ("pc", br + 1, rows[br].pc + 8))    # nor go elsewhere
    alu = next(r for r, w in enumerate(rows) if (w.insn & 0x7f) == 0x13 and r > 4)
    assert "next:plain" in B(("next_pc", alu, rows[alu].pc + 8), ("pc", alu + 1, rows[alu].pc + 8))  # an ALU instruction that skips the next one
    assert B(("opc_jal", alu, 1))                                             # ... and cannot be flagged as a jump to get away with it
    io = next(r for r, w in enumerate(rows) if w.insn == 0x73 and w.next_pc == w.pc)  # an I/O ecall repeating: to pc or pc + 4, nowhere else
    assert "next:ecall" in B(("next_pc", io, rows[io].pc + 8), ("pc", io + 1, rows[io].pc + 8))

    # ---- memory consistency: registers, memory and instruction words
    def next_read(reg, after):
        return next(r for r in range(after + 1, n) if ((rows[r].insn >> 15) & 31) == reg or ((rows[r].insn >> 20) & 31) == reg)

    w_row = next(r for r, w in enumerate(rows) if w.rd and r > 8 and any(((q.insn >> 15) & 31) == w.rd for q in rows[r + 1:r + 30]))
    reg = rows[w_row].rd
    r_row = next_read(reg, w_row)
    slot = "rs1" if ((rows[r_row].insn >> 15) & 31) == reg else "rs2"
    v = getattr(rows[r_row], slot + "_value")
    mem_sum = lambda names: any(x.startswith("sum:") for x in names)
    assert mem_sum(B((slot + "_lo", r_row, (v & 0xffff) ^ 1)))                 # a register that changes between its write and the next read
    assert mem_sum(B(("old_hi", w_row, (rows[w_row].rd_before >> 16) ^ 1)))    # a write that misstates what it overwrote
    st = next(r for r, w in enumerate(rows) if w.mem_kind == r0.MEM_WRITE and r > 20)
    assert B(("after_lo", st, (rows[st].mem_after & 0xffff) ^ 4))              # a store whose word is not what the instruction stores
    assert mem_sum(B(("dl4", mid, 5 * mid + 1 - rows[mid].prev[4] - 2)))       # a made-up previous timestamp (the consumed one IS own - 1 - the limbs)
    assert "lookup dh4" in B(("dl4", mid, P - 6 & 0xffff), ("dh4", mid, (P - 6) >> 16))  # a tuple from the future: the difference is no 24-bit number
    ld = next(r for r, w in enumerate(rows) if w.mem_kind == r0.MEM_READ and (w.insn & 0x7f) == 0x03 and r > 20)
    assert "ordered:mem" in B(("p3", ld, rows[ld].prev[3] + 1))                # (access 3 keeps a column for it: tied to its limbs)
    x0 = next(r for r, w in enumerate(rows) if ((w.insn >> 15) & 31) == 0 and (w.insn & 0x7f) == 0x13)
    assert mem_sum(B(("rs1_lo", x0, 5)))                                       # x0 reads what was last written there: nothing ever is
    bounds = vm.boundary(0)
    b0 = n + 3
    assert mem_sum(B(("after_lo", b0, (bounds[3].first_value & 0xffff) ^ 1), ("ob0", b0, (bounds[3].first_value & 1) ^ 1)))  # the first value of an address is what its first access finds
    assert mem_sum(B(("p3", b0, bounds[3].last_ts + 5)))
    assert "bnd:order" in B(("addr3", b0, bounds[2].addr), ("dl0", b0, bounds[2].addr & 0xffff), ("dl1", b0, bounds[2].addr >> 16))  # an address twice among the boundary rows (two histories)
    assert B(("bnd", b0, 0))                                                   # a boundary row dropped
    # a consistent lie about what an instruction computed -- the value written changed with every later sight of it, up to the register's
    # next write or its boundary row: memory stays consistent, the instruction does not
    lie = (rows[w_row].rd_after & 0xffff) ^ 1
    chain = [("res_lo", w_row, lie)]
    r = w_row
    while True:
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
    got = B(*chain)
    assert got and not any(x.startswith("sum:rd") for x in got), got


def forged_result(r, value, word="z"):
    """the edits of a prover that claims instruction r wrote `value`: the result columns (what the register receives) and the
    range-checked word the result is read from (Z or W) with everything the always-on definitions derive from it"""
    lo, hi = value & 0xFFFF, value >> 16
    edits = [("res_lo", r, lo), ("res_hi", r, hi)]
    if word == "u":
        return edits + [("u%d" % i, r, (value >> (8 * i)) & 255) for i in range(4)] + [("su", r, value >> 31)]
    edits += [(word + "_hi", r, hi)] + ([("w_lo", r, lo)] if word == "w" else [])  # (Z's low half is no column: its two low bits and the rest)
    if word == "z":
        edits += [("ob0", r, value & 1), ("ob1", r, (value >> 1) & 1), ("zq", r, lo >> 2), ("eq", r, int(value == 0)), ("zinv", r, pow(lo + hi, P - 2, P) if value else 0)]
    return edits


def test_what_an_instruction_computes_is_constrained_kind_by_kind(orc):
    """Random programs over every RV32IM instruction kind (tools/soak_trace.py): the genuine witness satisfies every constraint and
    fraction, and for each kind that writes a register the most careful lie available -- another value written, the result columns
    and the range-checked word changed with it -- breaks a constraint that belongs to that instruction's unit.  Stores: another
    word written.  Branches: the other way taken."""
    from soak_trace import random_program
    rng = np.random.default_rng(21)
    seen = {}
    for trial in range(4):
        vm = r0.Vm()
        vm.load(0x1000, random_program(rng, 350))
        vm.set_pc(0x1000)
        for i in range(1, 28):
            vm.set_reg(i, int(rng.integers(0, 1 << 32)) if rng.random() < 0.7 else int(rng.choice([0, 1, 0xFFFFFFFF, 0x80000000])))
        vm.set_input([int(x) for x in rng.integers(0, 1 << 32, 8)])
        assert vm.run(segment_po2=20, keep_trace=True, boundary_rows=True, max_cycles=50_000) == (0, 0)
        data, glob = vm.trace_witness(0, PO2)
        m0, g = canonical(data, PO2), [int(x) * R_INV % P for x in glob]
        assert check_trace_rows(m0, g) == []
        rows = vm.preflight(0)
        # every kind's lie in ONE altered witness per program (the rows are far apart; a constraint names the rows it fails on)
        m, forged = m0.copy(), {}
        for r, w in enumerate(rows):
            op, f3, f7 = w.insn & 0x7F, (w.insn >> 12) & 7, w.insn >> 25
            kind = (op, f3, f7 if op == 0x33 else (f7 & 0x20) if (op == 0x13 and f3 == 5) else 0)
            if kind in seen or kind in forged.values() or r == len(rows) - 1 or r < 2 or any(abs(r - q) < 3 for q in forged):
                continue
            if w.rd and op != 0x73:
                from_w = op in (0x6F, 0x67) or (op in (0x13, 0x33) and f3 == 5 and f7 != 1) or (op == 0x33 and f7 == 1 and f3 in (1, 2, 3))
                from_u = op == 0x33 and f7 == 1 and f3 in (4, 5)  # a quotient
                edits = forged_result(r, w.rd_after ^ 0x10, "u" if from_u else "w" if from_w else "z")
            elif op == 0x23:
                edits = [("after_lo", r, (w.mem_after & 0xFFFF) ^ 0x100)]
            elif op == 0x63:
                other = w.pc + 4 if w.next_pc != w.pc + 4 else (w.pc + 8) & 0xFFFFFFFF
                edits = [("next_pc", r, other), ("pc", r + 1, other)]
            else:
                continue
            for c, rr, v in edits:
                m[COL[c], rr] = v % P
            forged[r] = kind
        try:
            tc.multiplicities(m, g)
        except ValueError:
            pass  # (a forged half that leaves its table shows up below as an open lookup)
        bad = check_trace_rows(m, g, first_only=False)
        for r, kind in forged.items():
            mine = [name for name, where in bad if r in where and not name.startswith("sum:")]
            op = kind[0]
            assert mine, (hex(rows[r].insn), kind)
            if op == 0x23:
                assert all(name.startswith(("sb:", "sh:", "sw:")) for name in mine), (hex(rows[r].insn), mine)
            elif op == 0x63:
                assert "next:branch" in mine
            else:
                assert not any(name.startswith(("run:", "bit:", "digit:")) for name in mine), (hex(rows[r].insn), mine)
            seen[kind] = mine
    ops = {k[0] for k in seen}
    assert {0x37, 0x17, 0x6F, 0x67, 0x63, 0x03, 0x23, 0x13, 0x33} <= ops and len(seen) >= 45, sorted(seen)
    assert {(0x33, f3, 1) for f3 in range(8)} <= set(seen) and {(0x33, 0, 0x20), (0x33, 5, 0x20), (0x13, 5, 0x20), (0x13, 1, 0)} <= set(seen)
    assert {(0x03, f3, 0) for f3 in (0, 1, 2, 4, 5)} | {(0x23, f3, 0) for f3 in range(3)} | {(0x63, f3, 0) for f3 in (0, 1, 4, 5, 6, 7)} <= set(seen)


def test_an_ecall_row_does_what_its_function_says(orc):
    """READ_WORDS / COMMIT / CYCLES / HALT rows: a7 and a0 are the two registers read, the transfers count a1 down and touch the
    word a0 + 4 (a1 - 1), HALT touches nothing.  The words READ_WORDS stores and the value CYCLES writes are the host's to say (any
    32-bit word); the words COMMIT reads are named in the session sum (test_the_session_binds_...)."""
    from test_rv32im import ADDI, A0, A1, A7, ECALL, LI, flat
    buf = r0.JOURNAL_BASE
    prog = flat(LI(A0, buf), ADDI(A1, 0, 3), ADDI(A7, 0, 1), ECALL, ADDI(A1, 0, 2), ADDI(A7, 0, 2), ECALL, ADDI(A7, 0, 3), ECALL,
                LI(A0, buf), ADDI(A1, 0, 0), ADDI(A7, 0, 1), ECALL, ADDI(A0, 0, 5), ADDI(A7, 0, 0), ECALL)
    vm = r0.Vm()
    vm.load(0x1000, prog)
    vm.set_pc(0x1000)
    vm.set_input([11, 22, 33])
    assert vm.run(segment_po2=20, keep_trace=True, boundary_rows=True) == (0, 5) and vm.journal == struct.pack("<II", 11, 22)
    assert broken(vm, 0) == []
    rows = vm.preflight(0)
    sysrows = [(r, w) for r, w in enumerate(rows) if w.insn == 0x73]
    assert [w.rs1_value for _, w in sysrows] == [1, 1, 1, 1, 2, 2, 2, 3, 1, 0]
    B = lambda *e: broken(vm, 0, PO2, list(e))
    rd0 = sysrows[0][0]      # READ_WORDS, a1 = 3: writes word 2 of the buffer
    assert rows[rd0].mem_addr == buf + 8 and rows[rd0].mem_after == 33
    assert "ecall:addr" in B(("addr3", rd0, (buf + 4) >> 2))                                                   # another word of the buffer
    assert "ecall:count_lo" in B(*forged_result(rd0, 1))                                                        # a1 skips a step
    assert "next:ecall" in B(("next_pc", rd0, rows[rd0].pc + 4), ("pc", rd0 + 1, rows[rd0].pc + 4))
    assert "ecall:mem_wr" in B(("mem_wr", rd0, 0))                                                             # READ_WORDS writes memory
    word = [("after_lo", rd0, 0x1234), ("after_hi", rd0, 0x5678), ("w_lo", rd0, 0x1234), ("w_hi", rd0, 0x5678)]
    got = B(*word)
    assert got and all(x.startswith("sum:mem") for x in got)                                                    # the word read in is the host's to choose: no constraint objects,
    assert "lookup w_hi" in B(*(word[:3] + [("after_hi", rd0, 0x15678), ("w_hi", rd0, 0x15678)]))              # only the later sights of it would -- as long as it is a word
    cm = sysrows[4][0]       # COMMIT, a1 = 2: reads word 1, an active commit
    assert rows[cm].mem_kind == r0.MEM_READ and "ecall:commit" in B(("cact", cm, 0))                           # ... which the session sum must hear of
    done = sysrows[3][0]     # a1 = 0: falls through, touches nothing
    assert (rows[done].mem_kind, rows[done].next_pc, rows[done].rd, rows[done].rd_after) == (0, rows[done].pc + 4, A1, 0)
    assert "ecall:mem" in B(("cact", done, 1))
    cyc = sysrows[7][0]
    other = (123456 << 2) | (rows[cyc].rd_after & 3)  # (Z's two low bits also select a byte of U; kept, so nothing else has to follow)
    got = B(("res_lo", cyc, other & 0xFFFF), ("res_hi", cyc, other >> 16), ("z_hi", cyc, other >> 16), ("zq", cyc, (other & 0xFFFF) >> 2))
    assert rows[cyc].rd == A0 and all(x.startswith("sum:") for x in got)                                       # CYCLES: any word (only a later read of a0 would tell)
    assert "ecall:cycles" in B(("fn_cyc", cyc, 0))                                                             # ... into a0: the register the tuple names follows the function
    halt = sysrows[-1][0]
    assert "ecall:mem" in B(("mem_wr", halt, 1))                                                               # HALT does not touch memory
    assert "ecall:act2" in B(("act2", halt, 1))                                                                # ... nor a register
    # the public inputs say how the segment ends: HALT with exit code 5
    data, glob = vm.trace_witness(0, PO2)
    g = [int(x) * R_INV % P for x in glob]
    assert g[8:18] == [0x1000, rows[-1].next_pc, len(rows), 1, 1, 5, 0, 1, 1, 0]
    m = canonical(data, PO2)
    for k, wrong, name in ((11, 2, "exit:end_kind"), (13, 6, "exit:end_lo"), (14, 1, "exit:end_hi"), (12, 0, "exit:end_is")):
        bad = list(g)
        bad[k] = wrong
        assert name in [nm for nm, _ in check_trace_rows(m, bad)], name
    cut = list(g)
    cut[11:15] = [0, 0, 0, 0]                                                                                 # "this segment was merely cut"
    assert "exit:end_is" in [nm for nm, _ in check_trace_rows(m, cut)]
    # ... and a HALT is the last cycle of its segment: nothing runs after it
    assert "exit:last_cycle" in B(("live", len(rows), 1))
    # an unknown function number has no satisfying row (the executor traps on it)
    assert "ecall:fn_max" in B(("rs1_lo", halt, 5), ("f0", halt, 1), ("f2", halt, 1), ("u0", halt, 5), ("a0", halt, 5 & rows[halt].rs2_value))


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
    assert broken(vm, 0) == []
    rows = vm.preflight(0)
    data, glob = vm.trace_witness(0, PO2)
    m0, g = canonical(data, PO2), [int(x) * R_INV % P for x in glob]
    M32 = 0xFFFFFFFF
    m, forged = m0.copy(), []
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
        assert sum(int(m0[COL["u%d" % i], r]) << (8 * i) for i in range(4)) == q                               # U: the quotient
        assert int(m0[COL["ob0"], r]) + 2 * int(m0[COL["ob1"], r]) + 4 * int(m0[COL["zq"], r]) + (int(m0[COL["z_hi"], r]) << 16) == rem  # Z: the remainder
        if b_ == 0 or (signed and a_ == 0x80000000 and b_ == M32):
            continue
        # the neighbouring solution of the division identity: quotient + 1, remainder - divisor (the identity's own columns are edited
        # and the comparison must object)
        q2, rem2 = (q + 1) & M32, (rem - b_) & M32
        for i in range(4):
            m[COL["u%d" % i], r] = (q2 >> (8 * i)) & 255
            m[COL["a%d" % i], r] = ((q2 >> (8 * i)) & 255) & ((b_ >> (8 * i)) & 255)
        m[COL["su"], r] = q2 >> 31
        m[COL["z_hi"], r], m[COL["zq"], r] = rem2 >> 16, (rem2 & 0xFFFF) >> 2
        m[COL["ob0"], r], m[COL["ob1"], r] = rem2 & 1, (rem2 >> 1) & 1
        m[COL["c1"], r] = rem2 >> 31
        m[COL["aux1"], r] = 2 * ((rem2 >> 16) & 0x7FFF)
        m[COL["sb"], r] = (q2 >> (8 * (rem2 & 3))) & 255
        lo, hi = (q2 if f3 in (4, 5) else rem2) & 0xFFFF, (q2 if f3 in (4, 5) else rem2) >> 16
        m[COL["res_lo"], r], m[COL["res_hi"], r] = lo, hi
        forged.append((r, a_, b_, f3))
    tc.multiplicities(m, g)
    bad = check_trace_rows(m, g, first_only=False)
    for r, a_, b_, f3 in forged:
        mine = [name for name, where in bad if r in where]
        assert any(name.startswith("div:") for name in mine), (a_, b_, f3, mine)
    assert len(forged) >= 60


def test_an_instruction_may_read_and_write_its_own_word(orc):
    """The fetch is the earliest access of a cycle (round 3's advisor: with the fetch stamped last, a load of the instruction's own word
    had no satisfying row): `lw` of the word being executed, and a store over it, both give consistent traces."""
    from test_rv32im import ADDI, A0, A7, ECALL, I, S, U, flat
    prog = flat(U(0, 6, 0x17),                  # auipc x6, 0
                I(4, 6, 2, 5, 0x03),            # lw x5, 4(x6): loads this very word
                ADDI(7, 0, 0x13),               # x7 = the word of `addi x0, x0, 0`
                U(0, 6, 0x17),                  # auipc x6, 0
                S(4, 7, 6, 2),                  # sw x7, 4(x6): overwrites itself (after it was fetched)
                ADDI(A0, 0, 0), ADDI(A7, 0, 0), ECALL)
    vm = r0.Vm()
    vm.load(0x1000, prog)
    vm.set_pc(0x1000)
    assert vm.run(segment_po2=20, keep_trace=True, boundary_rows=True) == (0, 0)
    rows = vm.preflight(0)
    assert rows[1].rd_after == prog[1] and rows[1].prev[3] == 5 * 1 + 1 and rows[4].mem_before == prog[4] and rows[4].prev[3] == 5 * 4 + 1  # the memory access follows the fetch of its own cycle
    assert vm.read(0x1010, 1)[0] == 0x13 and broken(vm, 0) == []


def _image(base, prog):
    return [((base >> 2) + i, w) for i, w in enumerate(prog)]


def test_the_session_binds_the_segments_the_program_and_the_journal(orc, prover):
    """Round 4: a receipt over the trace circuit says "THIS program produced THIS journal".  A run cut into two
    segments (the second one closes the session), proved by hand with the oracle: the receipt verifies with the program; the same
    receipt is refused with another program; a boundary value altered between two segments, a journal byte altered (output digest
    recomputed), another program's state planted as the first claim, a challenge of the prover's own choosing -- all refused."""
    from test_rv32im import ADDI, A0, A7, ECALL, flat
    from bench_session import elf_of  # the ELF the verifier is given
    blob, pv = prover
    prog, base = _guest(160), 0x400
    elf = elf_of(prog, base)

    def run():
        vm = r0.Vm()
        vm.load_elf(elf)
        vm.set_input([7, 0x01020304])
        assert vm.run(segment_po2=10, keep_trace=True, boundary_rows=True) == (0, 0)
        return vm

    vm = run()
    segs = vm.segments()
    assert len(segs) == 2 and [s.closing for s in segs] == [0, 1] and all(s.user_cycles for s in segs)
    # the tuples balance: what a segment finds in an address is what the previous holder left, the first holder finds the image or zero,
    # the journal words are the ones the COMMIT rows read (numpy, exact)
    extra = [(a, w & 0xFFFF, w >> 16, tc.TAG_IMG, -1) for a, w in _image(base, prog)]
    extra += [(r0.JOURNAL_BASE // 4 + i, w & 0xFFFF, w >> 16, tc.TAG_JRN, -1) for i, w in enumerate(struct.unpack("<2I", vm.journal))]
    code = tc.code_columns(1 << PO2)
    _, session = tc.fractions()
    net = {}
    for k in range(2):
        data, glob = vm.trace_witness(k, PO2)
        m, g = canonical(data, PO2), [int(x) * R_INV % P for x in glob]
        assert check_trace_rows(m, g) == []
        for f in session:
            num = f.num.evaluate(m, code, g)
            parts = [lf.evaluate(m, code, g) for _, lf in f.parts[1:]]
            for r in np.nonzero(num)[0]:
                key = tuple(int(p[r]) for p in parts)
                net[key] = (net.get(key, 0) + int(num[r])) % P
    for a, lo, hi, t, num in extra:
        key = ((-a) % P, (-lo) % P, (-hi) % P, (-t) % P)
        net[key] = (net.get(key, 0) + num) % P
    assert not any(net.values())
    # the same through proofs: the receipt verifies with the ELF, and only with it
    receipt, roots = sbh.prove_session(pv, vm)
    image_id = r0.compute_image_id(elf)
    assert receipt.verify(blob, roots, None, elf=elf)[:2] == (0, "ok")
    assert receipt.verify(blob, roots, image_id)[0] == 15                                       # the image id alone leaves the session sum unchecked: not OK
    other = elf_of(prog[:-4] + [ADDI(0, 0, 0)] + prog[-3:], base)                               # one instruction the run never reached differs
    assert receipt.verify(blob, roots, None, elf=other)[0] in (8, 14)
    seals = [s for _, s in receipt.seals()]
    assert all(np.array_equal(s[r0.TRACE_GAMMA:r0.TRACE_GAMMA + 16], seals[0][r0.TRACE_GAMMA:r0.TRACE_GAMMA + 16]) for s in seals)
    # (iii) a value altered on its way from segment 0 to segment 1: segment 1 "finds" another word than segment 0 left
    vm2 = run()
    b1 = vm2.boundary(1)
    j = next(i for i, b in enumerate(b1) if b.prev_seg == 1 and b.addr < REG and b.first_value == b.last_value)  # a word segment 1 only reads
    n1 = segs[1].user_cycles

    def alter(k, data, glob):
        if k != 1:
            return False
        N = 1 << PO2
        row = n1 + j
        v = b1[j].first_value ^ 0x40
        for col, val in (("after_lo", v & 0xFFFF), ("zq", (v & 0xFFFF) >> 2), ("before_lo", v & 0xFFFF)):
            data[COL[col] * N + row] = orc.enc(val)
        # ... and every read of it inside segment 1 sees the altered word (memory stays consistent within the segment)
        for r, w in enumerate(vm2.preflight(1)):
            if w.mem_kind and (w.mem_addr >> 2) == b1[j].addr:
                for col in ("before_lo", "after_lo"):
                    data[COL[col] * N + r] = orc.enc(v & 0xFFFF)
        return True

    forged, roots2 = sbh.prove_session(pv, vm2, edit=alter)
    verdict = forged.verify(blob, roots2, None, elf=elf)
    assert verdict[0] in (2, 14), verdict                                                       # the seal itself (a load's result no longer follows) or the session sum
    # (iv) a journal byte altered, the output digest recomputed: the COMMIT rows named another word
    vm3 = run()
    journal = bytearray(vm3.journal)
    journal[0] ^= 1
    claims = vm3.claims()
    claims[-1] = r0.ReceiptClaim.make(claims[-1].pre, claims[-1].post, 0, 0, output_digest=r0.output_digest(bytes(journal)))
    forged, roots3 = sbh.prove_session(pv, vm3, claims=claims, journal=bytes(journal))
    assert forged.verify(blob, roots3, None, elf=elf)[:2] == (14, "the segments' session sums do not balance with the program image and the journal")
    # (ii) program B executed, program A's state planted as the first claim (the seals name the forged claims: proved anew)
    prog_b = prog[:-3] + [ADDI(A0, 0, 0), ADDI(A7, 0, 0), ECALL]                                # same length; B == A here is fine for the run, the IMAGE below differs
    elf_a = elf_of([ADDI(0, 0, 1)] + prog[1:], base)                                            # "program A": its first word differs from what ran
    vm4 = run()
    claims = vm4.claims()
    vm_a = r0.Vm()
    vm_a.load_elf(elf_a)
    assert vm_a.run(max_cycles=1)[0] == r0.Vm.LIMIT
    pre_a = vm_a.segments()[0].pre                                                              # the state a run of A starts from
    claims[0] = r0.ReceiptClaim.make(pre_a, claims[0].post, 2, 0)
    forged, roots4 = sbh.prove_session(pv, vm4, claims=claims)
    assert r0.compute_image_id(elf_a) == bytes(pre_a.digest())
    v = forged.verify(blob, roots4, None, elf=elf_a)
    assert v[0] == 14, v                                                                        # the image id matches; the words the run fetched are not A's
    # a challenge of the prover's own choosing is not the session's
    forged, roots5 = sbh.prove_session(pv, run(), challenge=lambda rec: r0.session_challenge(rec[::-1].copy()))
    assert forged.verify(blob, roots5, None, elf=elf)[0] == 13


def test_the_image_proof_lets_the_image_id_alone_verify_a_session(orc, prover):
    """`receipt.verify(image_id)` as the reference calls it (verifier/src/main.rs:124-126): 32 bytes, no ELF.  The receipt carries an
    image proof (circuits/image.r0c): the Poseidon2 digest of the image's word list -- the root of the state the image id names -- is
    computed inside that proof, and its rows add the image words' fractions under the session's challenge; the verifier balances the
    session with the proof's total instead of walking the ELF.  Here the session and the image proof are made by the oracle; refused:
    a receipt without the proof, the proof of another image (whatever digest it claims), of another session (another challenge), a
    total that is not the image's, an image id that is not the first pre-state's."""
    from test_rv32im import ADDI
    from bench_session import elf_of
    import image_circuit as ic
    blob, pv = prover
    iblob = np.fromfile(circuit_path("image"), dtype=np.uint32)
    oi = orc.circuit(iblob)
    prog, base = _guest(120), 0x400
    elf = elf_of(prog, base)
    vm = r0.Vm()
    vm.load_elf(elf)
    vm.set_input([7, 0x01020304])
    assert vm.run(segment_po2=12, keep_trace=True, boundary_rows=True) == (0, 0)
    receipt, roots = sbh.prove_session(pv, vm)
    image_id = r0.compute_image_id(elf)
    assert receipt.verify(blob, roots, None, elf=elf)[:2] == (0, "ok") and receipt.verify(blob, roots, image_id)[0] == 15
    assert receipt.image_proof is None and receipt.verify_image(blob, roots, iblob, image_id)[0] == 16  # no image proof: not accepted on the id alone
    challenge = receipt.seals()[0][1][r0.TRACE_GAMMA:r0.TRACE_GAMMA + 16]

    def image_seal(of_elf, under=challenge, edit=None):
        po2 = r0.image_po2(of_elf)
        data, glob = r0.image_witness(of_elf, po2)
        glob[r0.IMAGE_GAMMA:r0.IMAGE_GAMMA + 16] = under
        code = oi.witgen(po2, 0)[0]
        glob = oi.logup_totals(po2, code, data.reshape(-1), glob)
        if edit is not None:
            edit(data, glob)
        try:
            return oi.prove(po2, code, data.reshape(-1), glob)
        except AssertionError:  # the oracle prover found no polynomial quotient
            return None

    # the library's witness is the generator's restatement, word for word, and its digest is the root the image id names
    po2 = r0.image_po2(elf)
    data, glob = r0.image_witness(elf, po2)
    cols, digest = ic.witness(_image(base, prog), 1 << po2)
    assert np.array_equal(np.array([[orc.enc(v) for v in c] for c in cols], dtype=np.uint32), data) and [orc.enc(v) for v in digest] == glob[:8].tolist()
    assert bytes(vm.segments()[0].pre.merkle_root) == b"".join(int(w).to_bytes(4, "little") for w in digest)
    for count in (1, 3, 4, 5, 8, 9):  # block boundaries: a short last block, a full one, one word over
        small = elf_of(prog[:count], base)
        d2, g2 = r0.image_witness(small, 9)
        c2, dg2 = ic.witness(_image(base, prog[:count]), 1 << 9)
        assert np.array_equal(np.array([[orc.enc(v) for v in col] for col in c2], dtype=np.uint32), d2) and [orc.enc(v) for v in dg2] == g2[:8].tolist(), count
        assert r0.image_po2(small) == 9
    good = image_seal(elf)
    assert oi.verify(good) == (0, "ok")
    receipt.image_proof = good
    assert receipt.verify_image(blob, roots, iblob, image_id)[:2] == (0, "ok")
    back = r0.Receipt.parse(receipt.to_json())  # the proof travels in the receipt's JSON
    assert np.array_equal(back.image_proof, good) and back.verify_image(blob, roots, iblob, image_id)[:2] == (0, "ok")
    assert back.verify_image(blob, roots, iblob, image_id, image_control_root=r0.control_root_host(iblob, po2))[:2] == (0, "ok")
    assert back.verify_image(blob, roots, iblob, image_id, image_control_root=r0.control_root_host(iblob, po2 + 1))[0] == 16
    assert receipt.verify_image(blob, roots, iblob, r0.compute_image_id(elf_of(prog, base + 4)))[0] == 8       # another image id
    # another program's image proof: honest about its own digest -> not this receipt's image
    other = elf_of(prog[:-4] + [ADDI(0, 0, 0)] + prog[-3:], base)
    receipt.image_proof = image_seal(other)
    assert receipt.verify_image(blob, roots, iblob, image_id)[0] == 16
    # ... claiming THIS image's digest over the other program's words: no such proof exists

    def claim_digest(data, glob):
        glob[:8] = r0.image_witness(elf, po2)[1][:8]
    forged = image_seal(other, edit=claim_digest)
    if forged is not None:
        assert oi.verify(forged)[0] != 0
        receipt.image_proof = forged
        assert receipt.verify_image(blob, roots, iblob, image_id)[0] == 16
    # the right image under another challenge (another session's proof replayed)
    receipt.image_proof = image_seal(elf, under=np.roll(challenge, 4))
    assert receipt.verify_image(blob, roots, iblob, image_id)[0] == 16
    # a total that is not the image's: a word's fraction left out (its flag cleared, the mask still hashed) -- or simply another number

    def drop_a_word(data, glob):
        data[r0.SPONGE_DATA_COLUMNS + 1, 0] = 0
        glob[:] = oi.logup_totals(po2, oi.witgen(po2, 0)[0], data.reshape(-1), glob)
    forged = image_seal(elf, edit=drop_a_word)
    assert forged is None or oi.verify(forged)[0] != 0

    def other_total(data, glob):
        glob[r0.IMAGE_SUM] = (int(glob[r0.IMAGE_SUM]) + 1) % P
    forged = image_seal(elf, edit=other_total)
    assert forged is None or oi.verify(forged)[0] != 0
    receipt.image_proof = good
    assert receipt.verify_image(blob, roots, iblob, image_id)[:2] == (0, "ok")


@pytest.mark.gpu
@pytest.mark.parametrize("n_loop,po2", [(60, 16), (9000, 17)])
def test_the_device_expands_and_proves_an_execution_trace_word_for_word_like_the_cpu_side(hal, orc, n_loop, po2):
    """The same on the GPU: the compact rows are uploaded and expanded by the device kernel (r0h_trace_witgen), the multiplicities
    counted on the device (r0h_logup_multiplicities) -- the DATA group equals the host reference's word for word (and the numpy
    restatement's in the columns that has); CODE columns (the lookup tables among them) generated on the device; the segment's sum,
    the accumulation and the seal equal the CPU port's and both verifiers accept, bound to the control root; a row that breaks the run
    is rejected."""
    vm, base = _run(n_loop)
    rows, bounds = vm.preflight_arrays(0)
    n = len(rows)
    assert (1 << (po2 - 2)) < n + len(bounds) <= (1 << po2) or po2 == r0.TRACE_MIN_PO2
    data, glob = vm.trace_witness(0, po2)
    blob = np.fromfile(circuit_path("trace"), dtype=np.uint32)
    c = orc.circuit(blob)
    gc = hal.load_circuit(blob)  # eval_check compiled in-process (hipRTC)
    code, synthetic, _ = hal.witgen(gc, po2, 0)
    synthetic.free()
    dev, dglob = hal.trace_witgen(rows, bounds, po2, number=1, closing=True, circuit=gc)
    assert np.array_equal(dglob, glob)
    got = dev.to_host()
    assert np.array_equal(got, data)
    if po2 <= 16:
        primary = [COL[c_] for c_ in PRIMARY]
        assert np.array_equal(got.reshape(r0.TRACE_COLUMNS, -1)[primary], montgomery(expand(rows, bounds, po2))[primary])
    ocode, _, _ = c.witgen(po2, 0)
    assert np.array_equal(ocode, code.to_host())
    rng = np.random.default_rng(po2)
    glob[r0.TRACE_GAMMA:r0.TRACE_GAMMA + 16] = [orc.enc(int(v)) for v in rng.integers(0, P, 16)]
    full = hal.logup_totals(gc, po2, code, dev, glob)
    assert np.array_equal(full, c.logup_totals(po2, ocode, data, glob))
    mix = np.array([orc.enc(int(v)) for v in rng.integers(0, P, c.n_mix)], dtype=np.uint32)
    acc = hal.accum_public(gc, po2, code, dev, full, mix)
    assert np.array_equal(acc.to_host(), c.accum_public(po2, ocode, data, full, mix))
    cc = hal.code_commit(gc, po2, code)
    seal = hal.prove_segment(gc, po2, cc, dev, full)
    root = cc.root()
    assert c.verify(seal, code_root=root) == (0, "ok") and r0.verify_seal(blob, seal, code_root=root)[:2] == (0, "ok")
    assert np.array_equal(seal, c.prove(po2, ocode, data, full))
    # a register value that changes between a write and the next read: rejected by both verifiers
    bad = rows.copy()
    k = next(r for r in range(20, n) if (bad[r, F["insn"]] >> 15) & 31)
    bad[k, F["rs1"]] ^= 1
    dev2, _ = hal.trace_witgen(bad, bounds, po2, circuit=gc)
    seal = hal.prove_segment(gc, po2, cc, dev2, hal.logup_totals(gc, po2, code, dev2, glob))
    assert c.verify(seal, code_root=root)[0] == 4 and r0.verify_seal(blob, seal, code_root=root)[0] == 4
    bad = rows.copy()
    bad[n // 2, F["pc"]] = 0x5000
    dev2.free()
    dev2, _ = hal.trace_witgen(bad, bounds, po2, circuit=gc)
    seal = hal.prove_segment(gc, po2, cc, dev2, hal.logup_totals(gc, po2, code, dev2, glob))
    assert c.verify(seal, code_root=root)[0] == 4 and r0.verify_seal(blob, seal, code_root=root)[0] == 4
    # a wrong result in a register nobody looks at before it is overwritten: memory stays consistent, the instruction does not
    # (a random program: the loop above reads everything it writes)
    from soak_trace import dead_write_lie, random_program
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
    dev2, glob2 = hal.trace_witgen(rows2, bounds2, po2, circuit=gc)
    glob2[r0.TRACE_GAMMA:r0.TRACE_GAMMA + 16] = glob[r0.TRACE_GAMMA:r0.TRACE_GAMMA + 16]
    assert c.verify(hal.prove_segment(gc, po2, cc, dev2, hal.logup_totals(gc, po2, code, dev2, glob2)), code_root=root) == (0, "ok")
    lie = dead_write_lie(rows2, bounds2, rng)
    assert lie is not None
    hal.trace_witgen(lie[0], lie[1], po2, into=dev2, circuit=gc)
    seal = hal.prove_segment(gc, po2, cc, dev2, hal.logup_totals(gc, po2, code, dev2, glob2))
    assert c.verify(seal, code_root=root)[0] == 4 and r0.verify_seal(blob, seal, code_root=root)[0] == 4
    with pytest.raises(r0.R0HipError, match="do not fit|outside"):
        hal.trace_witgen(rows, bounds, 15)
    cc.free(); code.free(); dev.free(); dev2.free(); acc.free(); gc.free()
