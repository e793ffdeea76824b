#!/usr/bin/env python3
"""Synthetic circuit blobs for the segment prover (format: include/r0hip_circuit.h).

The reference's real circuit (risc0-circuit-rv32im 4.0.4: tap table + generated constraint polynomial) is an
un-vendored, machine-generated artefact that cannot be reproduced here (hard part 2).  The prover takes
the circuit as DATA, so this tool emits circuits of the same *shape* -- three tap groups (ACCUM/CODE/DATA), registers
with back-offsets, a `PolyExtStep` program (Const/Get/GetGlobal/Add/Sub/Mul/True/AndEqz/AndCond), grand-product
accumulators gated by a first-row selector -- together with a column program from which a satisfying witness can be
generated, so that proofs of it verify.

Shapes:
  tiny   W=(8 accum, 4 code, 12 data)      unit tests (CPU oracle in milliseconds)
  small  W=(16, 8, 40)                      GPU parity tests
  bench  W=(48, 16, 192) = 256 columns      SURVEY.md 8(d) config 2 (20k mul + 30k add/sub per point, <= 600 taps)
  recursion  W=(24, 8, 96), 16 public inputs   the second circuit of SURVEY.md 8(a) a19 in shape only (lift/join at po2 = 18)
  trace  W=(16, 4, 144)                     columns = the executor's preflight rows; one contiguous run, control flow per the words, memory consistency
"""
import argparse
import struct
import sys

P = 15 * 2**27 + 1
MAGIC = 0x31433052
SEC_GROUPS, SEC_TAPS, SEC_GLOBALS, SEC_POLY, SEC_WITGEN, SEC_ACCUM, SEC_INFO = 1, 2, 3, 4, 5, 6, 7
G_ACCUM, G_CODE, G_DATA = 0, 1, 2
OP_CONST, OP_GET, OP_GET_GLOBAL, OP_ADD, OP_SUB, OP_MUL, OP_TRUE, OP_AND_EQZ, OP_AND_COND = 0, 2, 3, 4, 5, 6, 7, 8, 9
BETA = 11


class Rng:
    def __init__(self, seed):
        self.s = seed & (2**64 - 1)

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & (2**64 - 1)
        x = self.s
        x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & (2**64 - 1)
        x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & (2**64 - 1)
        return x ^ (x >> 31)

    def below(self, n):
        return self.next() % n

    def pick(self, seq):
        return seq[self.below(len(seq))]


def ref(group, col, back):
    return (group << 28) | (back << 20) | col


class Builder:
    """Collects taps symbolically; fp/mix variables are numbered in creation order (as risc0's PolyExtStepDef)."""

    def __init__(self):
        self.steps = []  # (op, a, b, c) with a = ('tap', g, col, back) for GET until finalised
        self.n_fp = 0
        self.n_mix = 0
        self.taps = set()
        self.memo = {}
        self.n_mul = 0
        self.n_add = 0

    def _fp(self, op, a=0, b=0, c=0):
        self.steps.append((op, a, b, c))
        self.n_fp += 1
        return self.n_fp - 1

    def _mix(self, op, a=0, b=0, c=0):
        self.steps.append((op, a, b, c))
        self.n_mix += 1
        return self.n_mix - 1

    def const(self, v):
        key = ("c", v % P)
        if key not in self.memo:
            self.memo[key] = self._fp(OP_CONST, v % P)
        return self.memo[key]

    def get(self, g, col, back):
        key = ("t", g, col, back)
        if key not in self.memo:
            self.taps.add((g, col, back))
            self.memo[key] = self._fp(OP_GET, key)
        return self.memo[key]

    def glob(self, base, off):
        key = ("g", base, off)
        if key not in self.memo:
            self.memo[key] = self._fp(OP_GET_GLOBAL, base, off)
        return self.memo[key]

    def add(self, a, b):
        self.n_add += 1
        return self._fp(OP_ADD, a, b)

    def sub(self, a, b):
        self.n_add += 1
        return self._fp(OP_SUB, a, b)

    def mul(self, a, b):
        self.n_mul += 1
        return self._fp(OP_MUL, a, b)

    def true(self):
        return self._mix(OP_TRUE)

    def and_eqz(self, x, v):
        self.n_mul += 4
        self.n_add += 4
        return self._mix(OP_AND_EQZ, x, v)

    def and_cond(self, x, cond, inner):
        self.n_mul += 20
        self.n_add += 16
        return self._mix(OP_AND_COND, x, cond, inner)


def fp4_mul_sym(b, x, y):
    """Schoolbook Fp4 product of two 4-lists of fp vars, x^4 = 11."""
    c = [None] * 7
    for i in range(4):
        for j in range(4):
            t = b.mul(x[i], y[j])
            c[i + j] = t if c[i + j] is None else b.add(c[i + j], t)
    beta = b.const(BETA)
    return [b.add(c[0], b.mul(beta, c[4])), b.add(c[1], b.mul(beta, c[5])), b.add(c[2], b.mul(beta, c[6])), c[3]]


def generate(n_code, n_data, n_acc, n_free, n_pad, n_global, seed, cond_every=5, comp=16):
    """Columns are grouped into components of `comp` DATA columns (as a real circuit's registers belong to one
    instruction/memory/... component): a derived column reads its own or the previous component, and a constraint
    touches the taps of one component plus CODE selectors.  Constraints are emitted component by component."""
    rng = Rng(seed)
    n_comp = (n_data + comp - 1) // comp
    free_per = max(1, n_free // n_comp)
    assert n_code >= 4 and n_data > n_free and free_per < comp
    code_cols = [(0, 0), (1, 0), (2, 0)] + [(3, k) for k in range(3, n_code)]
    is_free = [(k % comp) < free_per for k in range(n_data)]
    assert sum(is_free) >= max(2, n_global)
    data_cols = []
    for k in range(n_data):
        if is_free[k]:
            data_cols.append((0, 0, 0, 0, 0))
            continue
        lo = max(0, (k // comp - 1) * comp)  # start of the previous component

        def pick_ref():
            if rng.below(100) < 85:
                return ref(G_DATA, lo + rng.below(k - lo), rng.pick([0, 0, 0, 1, 1, 2]))
            return ref(G_CODE, rng.below(n_code), rng.pick([0, 0, 1]))

        kind = 2 if rng.below(3) == 0 else 1
        data_cols.append((kind, pick_ref(), pick_ref(), pick_ref() if kind == 2 else 0, pick_ref()))
    acc_cols = [(0, rng.below(n_data), rng.below(n_data)) for _ in range(n_acc)]
    global_cols = [k for k in range(n_data) if is_free[k]][:n_global]

    b = Builder()
    for g, size in ((G_ACCUM, 4 * n_acc), (G_CODE, n_code), (G_DATA, n_data)):
        for col in range(size):
            b.taps.add((g, col, 0))

    def rget(r):
        return b.get(r >> 28, r & 0xFFFFF, (r >> 20) & 0xFF)

    one = b.const(1)
    first = b.get(G_CODE, 0, 0)
    by_comp = [[] for _ in range(n_comp)]  # per component: (fp var that is zero on every trace row, degree)
    defects = [[] for _ in range(n_comp)]
    for k in range(n_data):
        if is_free[k]:
            continue
        kind, ra, rb, rc, re = data_cols[k]
        prod = b.mul(rget(ra), rget(rb))
        if kind == 2:
            prod = b.mul(prod, rget(rc))
        d = (b.sub(b.get(G_DATA, k, 0), b.add(prod, rget(re))), kind + 1)
        defects[k // comp].append(d)
        by_comp[k // comp].append(d)
    for q in range(n_comp):
        cols = range(q * comp, min(n_data, (q + 1) * comp))
        pool = sorted(t for t in b.taps if (t[0] == G_DATA and t[1] in cols)) + [(G_CODE, c, 0) for c in range(n_code)]
        share = n_pad // n_comp + (1 if q < n_pad % n_comp else 0)
        if not defects[q]:  # a trailing component of free columns only has nothing to pad
            share = 0
        for _ in range(share):
            d, deg = rng.pick(defects[q])
            t = [b.get(*rng.pick(pool)) for _ in range(7)]
            g = b.add(b.add(b.add(b.add(b.mul(t[0], t[1]), b.mul(t[2], t[3])), t[4]), t[5]), t[6])
            by_comp[q].append((b.mul(d, g), deg + 2))
    tail = [(b.mul(first, b.sub(b.get(G_DATA, global_cols[k], 0), b.glob(0, k))), 2) for k in range(n_global)]
    not_first = b.sub(one, first)
    for j, (_, ca, cb) in enumerate(acc_cols):
        a, bb = b.get(G_DATA, ca, 0), b.get(G_DATA, cb, 0)
        m0 = [b.glob(1, 8 * j + i) for i in range(4)]
        m1 = [b.glob(1, 8 * j + 4 + i) for i in range(4)]
        term = [b.add(m0[i], b.mul(m1[i], bb)) for i in range(4)]
        term[0] = b.add(term[0], a)
        prev = [b.get(G_ACCUM, 4 * j + i, 1) for i in range(4)]
        sel = [b.mul(not_first, prev[i]) for i in range(4)]
        sel[0] = b.add(sel[0], first)
        want = fp4_mul_sym(b, term, sel)
        tail.extend((b.sub(b.get(G_ACCUM, 4 * j + i, 0), want[i]), 3) for i in range(4))

    all_vals = [v for q in range(n_comp) for v in by_comp[q]] + tail
    assert max(deg for _, deg in all_vals) <= 5  # check = C / (x^N - 1) must stay below degree 4N
    # constraint chain; every cond_every-th run of 8 constraints sits inside an AndCond gated by a CODE column.
    # The gate costs one degree, so only runs of degree <= 4 may be gated.
    x = b.true()
    i, run = 0, 0
    while i < len(all_vals):
        chunk = all_vals[i:i + 8]
        i += 8
        run += 1
        if cond_every and run % cond_every == 0 and max(deg for _, deg in chunk) <= 4:
            inner = b.true()
            for v, _ in chunk:
                inner = b.and_eqz(inner, v)
            x = b.and_cond(x, b.get(G_CODE, 3 + rng.below(n_code - 3), 0), inner)
        else:
            for v, _ in chunk:
                x = b.and_eqz(x, v)
    ret = x

    taps = sorted(b.taps)
    tap_index = {t: i for i, t in enumerate(taps)}
    steps = []
    for op, a, bb, c in b.steps:
        if op == OP_GET:
            a = tap_index[(a[1], a[2], a[3])]
        steps.append((op, a, bb, c))

    def section(tag, words):
        return [tag, len(words)] + list(words)

    words = [MAGIC, 1, 7]
    words += section(SEC_INFO, list(struct.unpack("<4I", b"R0HIP_SYNTH:v1__")))
    words += section(SEC_GROUPS, [4 * n_acc, n_code, n_data])
    words += section(SEC_TAPS, [len(taps)] + [w for t in taps for w in t])
    words += section(SEC_GLOBALS, [n_global, 8 * n_acc] + global_cols)
    words += section(SEC_POLY, [len(steps), ret] + [w for s in steps for w in s])
    words += section(SEC_WITGEN, [n_code] + [w for cc in code_cols for w in cc] + [n_data] + [w for d in data_cols for w in d])
    words += section(SEC_ACCUM, [n_acc] + [w for a in acc_cols for w in a])
    info = {"taps": len(taps), "steps": len(steps), "constraints": len(all_vals), "mul_per_point": b.n_mul,
            "addsub_per_point": b.n_add, "groups": [4 * n_acc, n_code, n_data], "components": n_comp}
    return words, info


# ---- the trace circuit: columns ARE the executor's preflight rows (include/r0hip.h: r0h_preflight_row / r0h_preflight_bound;
# csrc/trace.hpp holds the same column list and the expansion of a row into it) ---------------------------------------------------
# Not the rv32im circuit (risc0's constrains every instruction's semantics; that tap table and polynomial are not reproducible
# here).  This one constrains
#   * that the cycles form ONE contiguous run from the public first pc to the public last pc in the public number of cycles;
#   * that control flow follows the instruction words (it leaves the sequential path only at JAL / JALR / branch / ecall words, JAL
#     and branches go where their immediates say, an ecall to pc or pc + 4);
#   * MEMORY CONSISTENCY over registers and memory as one address space, by offline memory checking: each of a cycle's five
#     accesses (x[rs1], x[rs2], x[rd], the memory word, the fetched word) reads the tuple (address, value, timestamp) that the
#     previous access to the address wrote and writes a new one with a larger timestamp (the difference is range-checked through
#     radix-4 digits); boundary rows, one per address in strictly increasing order, write the first tuple (timestamp 0) and read
#     the last.  Multiset equality of tuples read and written is a grand product in ACCUM (four running products over fingerprints
#     alpha - addr - b1 lo - b2 hi - b3 t with alpha, b1..b3 drawn after DATA is committed), compared on the last row.
# Public inputs: 8 words naming the segment's ReceiptClaim, first pc, pc after the last cycle, number of cycles.
TRACE_COLUMNS = (["live", "bnd", "cycle", "pc", "next_pc", "is_seq", "insn_lo", "insn_hi"]
                 + ["bit%d" % k for k in range(32)]                                                   # the instruction word, bit by bit
                 + ["is_jal", "is_jalr", "is_branch", "is_ecall", "inv_jal", "inv_jalr", "inv_branch", "inv_ecall"]  # opcode classes, pinned by inverses
                 + ["z1", "inv1", "act0", "addr0", "rs1_lo", "rs1_hi", "p0", "tw0"]                   # access 0: x[rs1] read
                 + ["z2", "inv2", "act1", "addr1", "rs2_lo", "rs2_hi", "p1", "tw1"]                   # access 1: x[rs2] read
                 + ["act2", "addr2", "inv_rd", "old_lo", "old_hi", "new_lo", "new_hi", "p2", "tw2"]  # access 2: x[rd] write
                 + ["mem_kind", "addr3", "before_lo", "before_hi", "after_lo", "after_hi", "p3", "tw3"]  # access 3: memory word / boundary row
                 + ["addr4", "p4", "tw4"]                                                             # access 4: instruction fetch
                 + ["d%d_%d" % (k, i) for k in range(5) for i in range(12)])                          # radix-4 digits of (own - previous - 1)
TRACE_GLOBALS = 11   # claim words 0..7, first pc, pc after the last cycle, cycles
REG_BASE = 1 << 30
SEC_ACCUM_FP = 8


def generate_trace():
    col = {name: i for i, name in enumerate(TRACE_COLUMNS)}
    n_data, n_code, n_acc, n_global = len(TRACE_COLUMNS), 4, 4, TRACE_GLOBALS
    code_cols = [(0, 0), (1, 0), (2, 0), (3, 3)]  # first-row indicator, last-row indicator, row index, one seeded column
    data_cols = [(0, 0, 0, 0, 0)] * n_data        # all free: the witness is the execution's
    b = Builder()
    for g, size in ((G_ACCUM, 4 * n_acc), (G_CODE, n_code), (G_DATA, n_data)):
        for c in range(size):
            b.taps.add((g, c, 0))

    def d(name, back=0):
        return b.get(G_DATA, col[name], back)

    one, two, three, four, five = (b.const(v) for v in (1, 2, 3, 4, 5))
    half = b.const((P + 1) // 2)
    first, last = b.get(G_CODE, 0, 0), b.get(G_CODE, 1, 0)
    not_first = b.sub(one, first)
    live, prev_live, bnd, prev_bnd = d("live"), d("live", 1), d("bnd"), d("bnd", 1)
    not_live = b.sub(one, live)
    cons = []  # (fp var that must vanish on every row, degree)

    def bit(v):
        cons.append((b.mul(v, b.sub(v, one)), 2))

    def lin(terms):  # sum of coeff * var, coeff an integer (negative allowed)
        acc = None
        for coeff, v in terms:
            t = v if coeff == 1 else b.mul(b.const(coeff % P), v)
            acc = t if acc is None else b.add(acc, t)
        return acc

    # --- the run: live rows first, then boundary rows, then blank rows
    bit(live)
    bit(bnd)
    cons.append((b.mul(live, bnd), 2))
    bit(d("is_seq"))
    cons.append((b.mul(b.mul(live, d("is_seq")), b.sub(d("next_pc"), b.add(d("pc"), four))), 3))        # sequential rows step by 4
    gate = b.mul(not_first, live)                                                                       # a live row that has a predecessor
    cons.append((b.mul(gate, b.sub(d("pc"), d("next_pc", 1))), 3))                                      # ... starts where that one went
    cons.append((b.mul(gate, b.sub(d("cycle"), b.add(d("cycle", 1), one))), 3))                         # ... one cycle later
    cons.append((b.mul(gate, b.sub(one, prev_live)), 3))                                                # ... and follows a live row
    cons.append((b.mul(b.mul(not_first, bnd), b.sub(one, b.add(prev_live, prev_bnd))), 3))              # a boundary row follows a live or a boundary row
    for name in ("pc", "next_pc", "cycle", "mem_kind", "act2"):                                         # rows that are not cycles carry none of these
        cons.append((b.mul(not_live, d(name)), 2))
    # --- the instruction word, its opcode class, control flow
    bits = [d("bit%d" % k) for k in range(32)]
    for bk in bits:
        bit(bk)
    cons.append((b.sub(d("insn_lo"), lin([(1 << k, bits[k]) for k in range(16)])), 1))
    cons.append((b.sub(d("insn_hi"), lin([(1 << k, bits[16 + k]) for k in range(16)])), 1))
    op = lin([(1 << k, bits[k]) for k in range(7)])
    flags = {}
    for name, code in (("jal", 0x6F), ("jalr", 0x67), ("branch", 0x63), ("ecall", 0x73)):
        f, inv_ = d("is_" + name), d("inv_" + name)
        diff = b.sub(op, b.const(code))
        bit(f)
        cons.append((b.mul(f, diff), 2))                                # flag = 1 only at this opcode ...
        cons.append((b.sub(b.mul(diff, inv_), b.sub(one, f)), 2))       # ... and 0 only elsewhere
        flags[name] = f
    jumpy = b.add(b.add(flags["jal"], flags["jalr"]), b.add(flags["branch"], flags["ecall"]))
    cons.append((b.mul(b.mul(live, b.sub(one, d("is_seq"))), b.sub(one, jumpy)), 3))
    step = b.sub(d("next_pc"), d("pc"))
    imm_j = lin([(-(1 << 20), bits[31])] + [(1 << k, bits[k]) for k in range(12, 20)] + [(1 << 11, bits[20])] + [(1 << (k - 20), bits[k]) for k in range(21, 31)])
    imm_b = lin([(-(1 << 12), bits[31]), (1 << 11, bits[7])] + [(1 << (k - 20), bits[k]) for k in range(25, 31)] + [(1 << (k - 7), bits[k]) for k in range(8, 12)])
    cons.append((b.mul(flags["jal"], b.sub(step, imm_j)), 2))
    cons.append((b.mul(b.mul(flags["branch"], b.sub(step, four)), b.sub(step, imm_b)), 3))
    cons.append((b.mul(b.mul(flags["ecall"], step), b.sub(step, four)), 3))                             # an I/O ecall repeats (pc) or completes (pc + 4)
    # --- the five accesses.  Timestamp of access k of a cycle: 5 cycle + k + 1.  An access that does not happen leaves its read
    # tuple equal to its written tuple (they cancel in the grand product); one that happens writes its own timestamp, larger than
    # the one it read: own - previous - 1 is a sum of twelve radix-4 digits.
    digit = [[d("d%d_%d" % (k, i)) for i in range(12)] for k in range(5)]
    for k in range(5):
        for dg in digit[k]:
            cons.append((b.mul(b.mul(dg, b.sub(dg, one)), b.mul(b.sub(dg, two), b.sub(dg, three))), 4))

    def stamp(k):
        return b.add(b.mul(five, d("cycle")), b.const(k + 1))

    def ordered(act, k, deg):  # act * (tw_k - p_k - 1 - digits_k) = 0
        cons.append((b.mul(act, b.sub(b.sub(d("tw%d" % k), b.add(d("p%d" % k), one)), lin([(4 ** i, digit[k][i]) for i in range(12)]))), deg + 1))

    reg_base = b.const(REG_BASE)
    for k, (z, inv_, act, lo_bit) in enumerate((("z1", "inv1", "act0", 15), ("z2", "inv2", "act1", 20))):
        idx = lin([(1 << i, bits[lo_bit + i]) for i in range(5)])
        cons.append((b.mul(d(z), idx), 2))                                                             # z = 1 iff the index is 0 ...
        cons.append((b.sub(b.mul(idx, d(inv_)), b.sub(one, d(z))), 2))
        cons.append((b.sub(d(act), b.mul(live, b.sub(one, d(z)))), 2))                                 # x0 is not memory: no access
        for half_ in ("lo", "hi"):
            cons.append((b.mul(d(z), d("rs%d_%s" % (k + 1, half_))), 2))                               # ... and reads as zero
        cons.append((b.mul(d(act), b.sub(d("addr%d" % k), b.add(reg_base, idx))), 2))                  # the register the word names
        cons.append((b.mul(d(act), b.sub(d("tw%d" % k), stamp(k))), 2))
        cons.append((b.mul(b.sub(one, d(act)), b.sub(d("tw%d" % k), d("p%d" % k))), 2))
        ordered(d(act), k, 1)
    act2 = d("act2")
    bit(act2)
    idx_rd = lin([(1 << i, bits[7 + i]) for i in range(5)])
    plain = b.mul(act2, b.sub(one, flags["ecall"]))                                                     # a register write of an ordinary instruction
    cons.append((b.mul(plain, b.sub(d("addr2"), b.add(reg_base, idx_rd))), 3))                          # ... goes to the register the word names,
    cons.append((b.mul(plain, b.sub(b.mul(idx_rd, d("inv_rd")), one)), 4))                              # ... which is not x0
    sys_wr = b.mul(act2, flags["ecall"])                                                                # an ecall writes a0 or a1
    cons.append((b.mul(b.mul(sys_wr, b.sub(d("addr2"), b.const(REG_BASE + 10))), b.sub(d("addr2"), b.const(REG_BASE + 11))), 4))
    cons.append((b.mul(act2, b.sub(d("tw2"), stamp(2))), 2))
    not_act2 = b.sub(one, act2)
    for a_, c_ in (("tw2", "p2"), ("new_lo", "old_lo"), ("new_hi", "old_hi")):
        cons.append((b.mul(not_act2, b.sub(d(a_), d(c_))), 2))
    ordered(act2, 2, 1)
    mk = d("mem_kind")
    cons.append((b.mul(b.mul(mk, b.sub(mk, one)), b.sub(mk, two)), 3))                                  # none / read / write
    mem_act = b.mul(half, b.mul(mk, b.sub(three, mk)))                                                  # 1 on reads and writes
    is_write = b.mul(half, b.mul(mk, b.sub(mk, one)))                                                   # 1 on writes
    keeps = b.sub(one, b.add(is_write, bnd))                                                            # the word stays as it was unless written (or a boundary row)
    cons.append((b.mul(keeps, b.sub(d("after_lo"), d("before_lo"))), 3))
    cons.append((b.mul(keeps, b.sub(d("after_hi"), d("before_hi"))), 3))
    cons.append((b.mul(mem_act, b.sub(d("tw3"), stamp(3))), 3))
    cons.append((b.mul(b.sub(one, b.add(mem_act, bnd)), b.sub(d("tw3"), d("p3"))), 3))
    cons.append((b.mul(bnd, d("tw3")), 2))                                                              # a boundary row writes the first tuple: timestamp 0
    ordered(mem_act, 3, 2)
    # boundary rows in strictly increasing address order: one history per address (16 digits: access 3's twelve, access 2's first four)
    gap = b.sub(b.sub(d("addr3"), b.add(d("addr3", 1), one)), lin([(4 ** i, digit[3][i]) for i in range(12)] + [(4 ** (12 + i), digit[2][i]) for i in range(4)]))
    cons.append((b.mul(b.mul(bnd, prev_bnd), gap), 3))
    cons.append((b.mul(live, b.sub(d("pc"), b.mul(four, d("addr4")))), 2))                              # the fetch reads the word at pc
    cons.append((b.mul(live, b.sub(d("tw4"), stamp(4))), 2))
    cons.append((b.mul(not_live, b.sub(d("tw4"), d("p4"))), 2))
    ordered(live, 4, 1)
    # --- public inputs: the run starts at pc0 in cycle 0; the row after the last cycle (or the last row itself) pins the end
    G0 = 8
    cons.append((b.mul(first, b.sub(live, one)), 2))
    cons.append((b.mul(first, b.sub(d("pc"), b.glob(0, G0))), 2))
    cons.append((b.mul(first, d("cycle")), 2))
    ended = b.mul(not_first, b.sub(prev_live, live))                                                    # 1 on the first row that is not a cycle
    cons.append((b.mul(ended, b.sub(d("next_pc", 1), b.glob(0, G0 + 1))), 3))
    cons.append((b.mul(ended, b.sub(b.add(d("cycle", 1), one), b.glob(0, G0 + 2))), 3))
    full = b.mul(last, live)                                                                            # a trace that fills every row
    cons.append((b.mul(full, b.sub(d("next_pc"), b.glob(0, G0 + 1))), 3))
    cons.append((b.mul(full, b.sub(b.add(d("cycle"), one), b.glob(0, G0 + 2))), 3))
    # --- the grand products: RS_A, RS_B over the tuples read, WS_A, WS_B over the tuples written
    alpha = [b.glob(1, i) for i in range(4)]
    beta = [[b.glob(1, 4 * (j + 1) + i) for i in range(4)] for j in range(3)]

    def fingerprint(addr, lo, hi, t):  # alpha - addr - b1 lo - b2 hi - b3 t as four base-field expressions
        out = []
        for i in range(4):
            e = b.sub(alpha[i], b.add(b.add(b.mul(beta[0][i], d(lo)), b.mul(beta[1][i], d(hi))), b.mul(beta[2][i], d(t))))
            out.append(b.sub(e, d(addr)) if i == 0 else e)
        return out

    products = [  # (ACCUM column block, the fingerprints it multiplies) -- mirrored in the SEC_ACCUM_FP records below
        [("addr0", "rs1_lo", "rs1_hi", "p0"), ("addr1", "rs2_lo", "rs2_hi", "p1"), ("addr2", "old_lo", "old_hi", "p2")],
        [("addr3", "before_lo", "before_hi", "p3"), ("addr4", "insn_lo", "insn_hi", "p4")],
        [("addr0", "rs1_lo", "rs1_hi", "tw0"), ("addr1", "rs2_lo", "rs2_hi", "tw1"), ("addr2", "new_lo", "new_hi", "tw2")],
        [("addr3", "after_lo", "after_hi", "tw3"), ("addr4", "insn_lo", "insn_hi", "tw4")],
    ]
    for j, tuples in enumerate(products):
        prev = [b.get(G_ACCUM, 4 * j + i, 1) for i in range(4)]
        want = [b.mul(not_first, prev[i]) for i in range(4)]
        want[0] = b.add(want[0], first)
        for t in tuples:
            want = fp4_mul_sym(b, want, fingerprint(*t))
        cons.extend((b.sub(b.get(G_ACCUM, 4 * j + i, 0), want[i]), 2 + len(tuples)) for i in range(4))
    acc = [[b.get(G_ACCUM, 4 * j + i, 0) for i in range(4)] for j in range(4)]
    reads, writes = fp4_mul_sym(b, acc[0], acc[1]), fp4_mul_sym(b, acc[2], acc[3])
    cons.extend((b.mul(last, b.sub(reads[i], writes[i])), 3) for i in range(4))                          # every tuple read was written, once
    assert max(deg for _, deg in cons) <= 5
    x = b.true()
    for v, _ in cons:
        x = b.and_eqz(x, v)
    taps = sorted(b.taps)
    tap_index = {t: i for i, t in enumerate(taps)}
    steps = [(op_, tap_index[(a_[1], a_[2], a_[3])] if op_ == OP_GET else a_, b_, c_) for op_, a_, b_, c_ in b.steps]

    def section(tag, words):
        return [tag, len(words)] + list(words)

    acc_records = []
    for tuples in products:
        rec = [len(tuples)]
        for t in tuples + [("live",) * 4] * (3 - len(tuples)):
            rec += [col[name] for name in t]
        acc_records += rec
    words = [MAGIC, 1, 7]
    words += section(SEC_INFO, list(struct.unpack("<4I", b"R0HIP_TRACE:v2__")))
    words += section(SEC_GROUPS, [4 * n_acc, n_code, n_data])
    words += section(SEC_TAPS, [len(taps)] + [w for t in taps for w in t])
    words += section(SEC_GLOBALS, [n_global, 16])
    words += section(SEC_POLY, [len(steps), x] + [w for st in steps for w in st])
    words += section(SEC_WITGEN, [n_code] + [w for cc in code_cols for w in cc] + [n_data] + [w for dc in data_cols for w in dc])
    words += section(SEC_ACCUM_FP, [n_acc] + acc_records)
    info = {"taps": len(taps), "steps": len(steps), "constraints": len(cons), "mul_per_point": b.n_mul, "addsub_per_point": b.n_add,
            "groups": [4 * n_acc, n_code, n_data], "columns": len(TRACE_COLUMNS)}
    return words, info


SHAPES = {
    "tiny": dict(n_code=4, n_data=12, n_acc=2, n_free=4, n_pad=6, n_global=2, seed=1, comp=6),
    # 8 public inputs: enough to name a receipt claim (r0h_claim_globals), as `bench` can
    "small": dict(n_code=8, n_data=40, n_acc=4, n_free=8, n_pad=60, n_global=8, seed=2, comp=10),
    "bench": dict(n_code=16, n_data=192, n_acc=12, n_free=24, n_pad=2600, n_global=8, seed=3, comp=16),
    # recursion-SHAPED: a smaller trace (proved at po2 = 18) whose 16 public inputs carry the two 8-word digests a lift/join step
    # stands for (hyperfridge-r0_amd/recursion.py).  It does not verify seals in-circuit: risc0's recursion circuit is not reproducible here.
    "recursion": dict(n_code=8, n_data=96, n_acc=6, n_free=20, n_pad=900, n_global=16, seed=4, comp=12),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("shape", choices=sorted(SHAPES) + ["trace"])
    ap.add_argument("out")
    args = ap.parse_args()
    words, info = generate_trace() if args.shape == "trace" else generate(**SHAPES[args.shape])
    with open(args.out, "wb") as f:
        f.write(struct.pack("<%dI" % len(words), *words))
    print(args.shape, info, "words", len(words))


if __name__ == "__main__":
    sys.exit(main())
