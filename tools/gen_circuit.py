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
  trace  W=(16, 4, 288)                     columns = the executor's preflight rows; one contiguous run, every instruction's semantics, memory consistency
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
# Stands where risc0's rv32im circuit stands; it is NOT that circuit (its tap table and polynomial are machine-generated and not
# reproducible here) but one written for this library's executor.  It constrains
#   * that the cycles form ONE contiguous run from the public first pc to the public last pc in the public number of cycles;
#   * WHAT EVERY INSTRUCTION DOES: the word is decoded (one-hot opcode and funct3, illegal encodings have no satisfying row); the
#     value written to rd, the word written to memory, the address of a load / store and the next pc are the ones RV32IM
#     prescribes for the operands read.  Units: a 32-bit adder over 16-bit halves (ADD / ADDI / AUIPC / address generation / JALR
#     target; run backwards for SUB, SLT[I][U] and the six branches), bit-sliced logic over the decomposed operands, a byte-limb
#     multiplier with a range-checked carry chain (MUL / MULH / MULHSU / MULHU, and the shifts as products with 2^s resp. 2^(32-s),
#     the sign of MULH* / SRA folded into the chain), byte / half selection for the narrow loads and stores.  Every word written
#     is range-checked (radix-4 digits) or composed of range-checked parts.  Not yet constrained: DIV / DIVU / REM / REMU results
#     and what an ecall row reads and writes (both are range-checked only) -- DESIGN.md 4;
#   * MEMORY CONSISTENCY over registers and memory as one address space, by offline memory checking: each of a cycle's five
#     accesses (x[rs1], x[rs2], x[rd], the memory word, the fetched word) reads the tuple (address, value, timestamp) that the
#     previous access to the address wrote and writes a new one with a larger timestamp (the difference is range-checked through
#     radix-4 digits); boundary rows, one per address in strictly increasing order (addresses below 2^28 + 32: 1 GiB of memory and
#     the registers above it), write the first tuple (timestamp 0) and read the last.  Multiset equality of tuples read and written is a grand product in ACCUM (four running products over fingerprints
#     alpha - addr - b1 lo - b2 hi - b3 t with alpha, b1..b3 drawn after DATA is committed), compared on the last row.
# Public inputs: 8 words naming the segment's ReceiptClaim, first pc, pc after the last cycle, number of cycles.
OPCODES = [("lui", 0x37), ("auipc", 0x17), ("jal", 0x6F), ("jalr", 0x67), ("branch", 0x63), ("load", 0x03), ("store", 0x23), ("imm", 0x13),
           ("op", 0x33), ("fence", 0x0F), ("system", 0x73)]
TRACE_COLUMNS = (["live", "bnd", "cycle", "pc", "next_pc", "insn_lo", "insn_hi"]
                 + ["bit%d" % k for k in range(32)]                                                   # the instruction word, bit by bit
                 + ["opc_" + name for name, _ in OPCODES]                                             # one-hot opcode
                 + ["f3_%d" % k for k in range(8)]                                                    # one-hot funct3
                 + ["alu"]                                                                            # OP-IMM or base-ISA OP (M-extension OP is opc_op * bit25)
                 + ["z1", "inv1", "act0", "addr0", "rs1_lo", "rs1_hi", "p0", "tw0"]                   # access 0: x[rs1] read
                 + ["z2", "inv2", "act1", "addr1", "rs2_lo", "rs2_hi", "p1", "tw1"]                   # access 1: x[rs2] read
                 + ["zrd", "inv_rd", "act2", "addr2", "old_lo", "old_hi", "new_lo", "new_hi", "p2", "tw2"]  # access 2: x[rd] write
                 + ["mem_kind", "addr3", "before_lo", "before_hi", "after_lo", "after_hi", "p3", "tw3"]  # access 3: memory word / boundary row
                 + ["addr4", "p4", "tw4"]                                                             # access 4: instruction fetch
                 + ["d%d_%d" % (k, i) for k in range(5) for i in range(12)]                           # radix-4 digits of (own - previous - 1)
                 + ["ub%d" % k for k in range(32)]                                                    # operand U bit by bit: x[rs1], or the memory word of a load / store
                 + ["vb%d" % k for k in range(32)]                                                    # operand V: x[rs2] or the I-immediate
                 + ["zd%d" % k for k in range(16)]                                                    # word Z in radix-4 digits: sum / difference / low product word
                 + ["wd%d" % k for k in range(16)]                                                    # word W: high product word, link, pc
                 + ["res_lo", "res_hi", "c0", "c1", "lt", "eq", "zinv", "ob0", "ob1", "sb", "sgn", "p8", "sx", "sm"]
                 + ["mb%d" % k for k in range(4)]                                                     # second multiplier operand, byte limbs
                 + ["cx%d" % k for k in range(4)] + ["c3"]                                            # carry digits beyond access 3's twelve; top carry
                 + ["dv", "ovf", "k0", "a31"] + ["at%d" % k for k in range(8)]                        # division: active, the overflow case, a carry, the dividend's sign and the digits under it
                 + ["io"])                                                                            # an ecall that moves words
TRACE_GLOBALS = 15   # claim words 0..7, first pc, pc after the last cycle, cycles, how the segment ends (0 cut / 1 HALT / 2 PAUSE), that being non-zero, exit code halves
REG_BASE = 1 << 28   # registers sit above 2^28 words = 1 GiB of memory
SEC_ACCUM_FP = 8


class E:
    """A base-field expression under construction: a Builder variable, its degree in the trace columns, its value if constant."""
    __slots__ = ("b", "v", "deg", "k")

    def __init__(self, b, v, deg, k=None):
        self.b, self.v, self.deg, self.k = b, v, deg, k

    @staticmethod
    def of(b, x):
        return x if isinstance(x, E) else E(b, b.const(x % P), 0, x % P)

    def __add__(self, o):
        o = E.of(self.b, o)
        if o.k == 0:
            return self
        if self.k == 0:
            return o
        if self.k is not None and o.k is not None:
            return E.of(self.b, self.k + o.k)
        return E(self.b, self.b.add(self.v, o.v), max(self.deg, o.deg))

    __radd__ = __add__

    def __sub__(self, o):
        o = E.of(self.b, o)
        if o.k == 0:
            return self
        if self.k is not None and o.k is not None:
            return E.of(self.b, self.k - o.k)
        return E(self.b, self.b.sub(self.v, o.v), max(self.deg, o.deg))

    def __rsub__(self, o):
        return E.of(self.b, o) - self

    def __mul__(self, o):
        o = E.of(self.b, o)
        if self.k is not None and o.k is not None:
            return E.of(self.b, self.k * o.k)
        for x, y in ((self, o), (o, self)):
            if x.k == 1:
                return y
            if x.k == 0:
                return x
        return E(self.b, self.b.mul(self.v, o.v), self.deg + o.deg)

    __rmul__ = __mul__


def lin(b, terms):
    acc = E.of(b, 0)
    for coeff, v in terms:
        acc = acc + coeff * v
    return acc


def trace_constraints():
    """-> (Builder, [(name, fp var, degree, touches ACCUM)], products): every polynomial that must vanish on every row of a trace"""
    col = {name: i for i, name in enumerate(TRACE_COLUMNS)}
    n_data, n_code, n_acc = len(TRACE_COLUMNS), 4, 4
    b = Builder()
    for g, size in ((G_ACCUM, 4 * n_acc), (G_CODE, n_code), (G_DATA, n_data)):
        for c in range(size):
            b.taps.add((g, c, 0))
    cons = []

    def d(name, back=0):
        return E(b, b.get(G_DATA, col[name], back), 1)

    def C(name, e, accum=False):
        assert e.deg <= 5, (name, e.deg)
        cons.append((name, e.v, e.deg, accum))

    def bit(v, name):
        C("bit:" + name, v * (v - 1))

    def digit(v, name):
        C("digit:" + name, v * (v - 1) * ((v - 2) * (v - 3)))

    first, last = E(b, b.get(G_CODE, 0, 0), 1), E(b, b.get(G_CODE, 1, 0), 1)
    not_first = 1 - first
    live, prev_live, bnd, prev_bnd = d("live"), d("live", 1), d("bnd"), d("bnd", 1)
    not_live = 1 - live
    pc, next_pc, cycle = d("pc"), d("next_pc"), d("cycle")
    # --- the run: live rows first, then boundary rows, then blank rows
    bit(live, "live")
    bit(bnd, "bnd")
    C("live.bnd", live * bnd)
    gate = not_first * live                                             # a live row that has a predecessor
    C("run:pc", gate * (pc - d("next_pc", 1)))                         # ... starts where that one went
    C("run:cycle", gate * (cycle - d("cycle", 1) - 1))                 # ... one cycle later
    C("run:after_live", gate * (1 - prev_live))                        # ... and follows a live row
    C("run:bnd_after", not_first * bnd * (1 - prev_live - prev_bnd))   # a boundary row follows a live or a boundary row
    for name in ("pc", "next_pc", "cycle", "mem_kind", "act2"):        # rows that are not cycles carry none of these
        C("idle:" + name, not_live * d(name))
    # --- decoding
    bits = [d("bit%d" % k) for k in range(32)]
    for k, bk in enumerate(bits):
        bit(bk, "bit%d" % k)
    C("insn_lo", d("insn_lo") - lin(b, [(1 << k, bits[k]) for k in range(16)]))
    C("insn_hi", d("insn_hi") - lin(b, [(1 << k, bits[16 + k]) for k in range(16)]))
    opc = {name: d("opc_" + name) for name, _ in OPCODES}
    for name, v in opc.items():
        bit(v, "opc_" + name)
    C("opc:one", lin(b, [(1, v) for v in opc.values()]) - live)
    C("opc:code", lin(b, [(code, opc[name]) for name, code in OPCODES]) - lin(b, [(1 << k, bits[k]) for k in range(7)]))
    f3 = [d("f3_%d" % k) for k in range(8)]
    for k, v in enumerate(f3):
        bit(v, "f3_%d" % k)
    C("f3:one", lin(b, [(1, v) for v in f3]) - 1)
    C("f3:code", lin(b, [(k, f3[k]) for k in range(8)]) - (bits[12] + 2 * bits[13] + 4 * bits[14]))
    alu, mext = d("alu"), opc["op"] * bits[25]
    C("alu", alu - (opc["imm"] + opc["op"] * (1 - bits[25])))
    for k in (26, 27, 28, 29, 31):
        C("op:f7_%d" % k, opc["op"] * bits[k])
    C("op:f7_m_alt", opc["op"] * bits[25] * bits[30])
    C("op:f7_alt", opc["op"] * bits[30] * (1 - f3[0] - f3[5]))
    for k in range(25, 32):
        C("slli:f7_%d" % k, opc["imm"] * f3[1] * bits[k])
        if k != 30:
            C("srxi:f7_%d" % k, opc["imm"] * f3[5] * bits[k])
    C("jalr:f3", opc["jalr"] * (1 - f3[0]))
    C("branch:f3", opc["branch"] * (f3[2] + f3[3]))
    C("load:f3", opc["load"] * (f3[3] + f3[6] + f3[7]))
    C("store:f3", opc["store"] * (1 - f3[0] - f3[1] - f3[2]))
    C("system:lo", opc["system"] * (d("insn_lo") - 0x73))              # ecall is the one SYSTEM word that runs
    C("system:hi", opc["system"] * d("insn_hi"))
    sign = bits[31]
    immi = [lin(b, [(1 << (k - 20), bits[k]) for k in range(20, 31)]) + 0xF800 * sign, 0xFFFF * sign]   # sign-extended, as two halves
    imms = [lin(b, [(1 << (k - 7), bits[k]) for k in range(7, 12)] + [(1 << (k - 20), bits[k]) for k in range(25, 31)]) + 0xF800 * sign, 0xFFFF * sign]
    immu = [lin(b, [(1 << k, bits[k]) for k in range(12, 16)]), d("insn_hi")]
    imm_j = lin(b, [(-(1 << 20), bits[31])] + [(1 << k, bits[k]) for k in range(12, 20)] + [(1 << 11, bits[20])] + [(1 << (k - 20), bits[k]) for k in range(21, 31)])
    imm_b = lin(b, [(-(1 << 12), bits[31]), (1 << 11, bits[7])] + [(1 << (k - 20), bits[k]) for k in range(25, 31)] + [(1 << (k - 7), bits[k]) for k in range(8, 12)])
    # --- the words the units work on
    ub, vb = [d("ub%d" % k) for k in range(32)], [d("vb%d" % k) for k in range(32)]
    for k in range(32):
        bit(ub[k], "ub%d" % k)
        bit(vb[k], "vb%d" % k)
    zd, wd = [d("zd%d" % k) for k in range(16)], [d("wd%d" % k) for k in range(16)]
    for k in range(16):
        digit(zd[k], "zd%d" % k)
        digit(wd[k], "wd%d" % k)
    halves = lambda digs: [lin(b, [(4 ** i, digs[i]) for i in range(8)]), lin(b, [(4 ** i, digs[8 + i]) for i in range(8)])]
    u = [lin(b, [(1 << k, ub[k]) for k in range(16)]), lin(b, [(1 << k, ub[16 + k]) for k in range(16)])]
    v = [lin(b, [(1 << k, vb[k]) for k in range(16)]), lin(b, [(1 << k, vb[16 + k]) for k in range(16)])]
    z, w = halves(zd), halves(wd)
    a = [d("rs1_lo"), d("rs1_hi")]
    rs2 = [d("rs2_lo"), d("rs2_hi")]
    before, after = [d("before_lo"), d("before_hi")], [d("after_lo"), d("after_hi")]
    res = [d("res_lo"), d("res_hi")]
    is_mem = opc["load"] + opc["store"]
    isdiv = mext * bits[14]
    sys_ = opc["system"]
    use_b, use_i = opc["op"] + opc["branch"] + opc["store"] + sys_, opc["imm"] + opc["load"] + opc["jalr"]
    for h, nm in enumerate(("lo", "hi")):
        C("u:" + nm, (1 - isdiv) * (u[h] - (is_mem * before[h] + (1 - is_mem) * a[h])))  # U: the word of a load / store, x[rs1] otherwise (a division row keeps its quotient there)
        C("v:" + nm, v[h] - (use_b * rs2[h] + use_i * immi[h]))               # V: x[rs2] or the I-immediate
    c0, c1, lt, eq, zinv, ob0, ob1 = d("c0"), d("c1"), d("lt"), d("eq"), d("zinv"), d("ob0"), d("ob1")
    for nm, x in (("c0", c0), ("c1", c1), ("lt", lt), ("eq", eq), ("ob0", ob0), ("ob1", ob1)):
        bit(x, nm)
    C("z:low_bits", zd[0] - ob0 - 2 * ob1)
    old = [d("old_lo"), d("old_hi")]
    # the zero test looks at Z; on a division row at the divisor; on an ecall row at the register it counts down
    zero_of = z[0] + z[1] + isdiv * (v[0] + v[1] - z[0] - z[1]) + sys_ * (old[0] + old[1] - z[0] - z[1])
    C("eq:zero", eq * zero_of)                                          # eq = 1 iff that word is 0 (both halves are 16-bit: no wrap)
    C("eq:inv", zero_of * zinv - (1 - eq))
    differ = ub[31] + vb[31] - 2 * ub[31] * vb[31]
    C("lt", lt - (differ * ub[31] + (1 - differ) * c1))                 # signed U < V given the borrow c1 of U - V
    # --- the adder: X + Y = Z + 2^32 carry (halves, two carry bits), or backwards: Y + Z = X + 2^32 borrow
    sub_rr = alu * f3[0] * (opc["op"] * bits[30])
    sel_add = opc["jalr"] + opc["load"] + alu * f3[0] - sub_rr
    sel_sub = opc["branch"] + alu * (f3[2] + f3[3]) + sub_rr
    C("add:lo", sel_add * (a[0] + v[0] - z[0] - 65536 * c0))
    C("add:hi", sel_add * (a[1] + v[1] + c0 - z[1] - 65536 * c1))
    C("store:addr_lo", opc["store"] * (a[0] + imms[0] - z[0] - 65536 * c0))
    C("store:addr_hi", opc["store"] * (a[1] + imms[1] + c0 - z[1] - 65536 * c1))
    C("auipc:pc", opc["auipc"] * (w[0] + 65536 * w[1] - pc))
    C("auipc:range", opc["auipc"] * wd[15])                             # W is the pc itself, not pc + p
    C("auipc:lo", opc["auipc"] * (w[0] + immu[0] - z[0] - 65536 * c0))
    C("auipc:hi", opc["auipc"] * (w[1] + immu[1] + c0 - z[1] - 65536 * c1))
    C("sub:lo", sel_sub * (v[0] + z[0] - a[0] - 65536 * c0))
    C("sub:hi", sel_sub * (v[1] + z[1] + c0 - a[1] - 65536 * c1))
    # --- control flow
    link = opc["jal"] + opc["jalr"]
    C("next:plain", (live - link - opc["branch"] - opc["system"]) * (next_pc - pc - 4))
    C("next:jal", opc["jal"] * (next_pc - pc - imm_j))
    C("next:jalr", opc["jalr"] * (next_pc - (z[0] + 65536 * z[1] - ob0)))
    C("jalr:range", opc["jalr"] * zd[15])                               # targets stay below 2^30: the pc is a field element
    C("jalr:aligned", opc["jalr"] * ob1)
    taken = f3[0] * eq + f3[1] * (1 - eq) + f3[4] * lt + f3[5] * (1 - lt) + f3[6] * c1 + f3[7] * (1 - c1)
    C("next:branch", opc["branch"] * (next_pc - pc - 4 - taken * (imm_b - 4)))
    step = next_pc - pc
    C("link", link * (w[0] + 65536 * w[1] - pc - 4))
    C("link:range", link * wd[15])
    # --- loads and stores
    C("mem:kind", (1 - sys_) * (d("mem_kind") - opc["load"] - 2 * opc["store"]))
    C("mem:addr", is_mem * (4 * d("addr3") + zd[0] - z[0] - 65536 * z[1]))
    C("mem:range", is_mem * zd[15])                                     # 1 GiB of memory
    narrow_h = opc["load"] * (f3[1] + f3[5]) + opc["store"] * f3[1]
    word = (opc["load"] + opc["store"]) * f3[2]
    C("mem:aligned_h", narrow_h * ob0)
    C("mem:aligned_w", word * (ob0 + ob1))
    sel_byte = [(1 - ob0) * (1 - ob1), ob0 * (1 - ob1), (1 - ob0) * ob1, ob0 * ob1]
    ubyte = [lin(b, [(1 << i, ub[8 * k + i]) for i in range(8)]) for k in range(4)]
    vbyte = [lin(b, [(1 << i, vb[8 * k + i]) for i in range(8)]) for k in range(4)]
    sb, sgn = d("sb"), d("sgn")
    C("sb", sb - lin(b, [(1, sel_byte[k] * ubyte[k]) for k in range(4)]))
    C("sgn", sgn - lin(b, [(1, sel_byte[k] * ub[8 * k + 7]) for k in range(4)]))
    sh = (1 - ob1) * u[0] + ob1 * u[1]
    sgnh = (1 - ob1) * ub[15] + ob1 * ub[31]
    ld = opc["load"]
    C("lb:lo", ld * f3[0] * (res[0] - sb - 0xFF00 * sgn))
    C("lb:hi", ld * f3[0] * (res[1] - 0xFFFF * sgn))
    C("lh:lo", ld * f3[1] * (res[0] - sh))
    C("lh:hi", ld * f3[1] * (res[1] - 0xFFFF * sgnh))
    C("lw:lo", ld * f3[2] * (res[0] - u[0]))
    C("lw:hi", ld * f3[2] * (res[1] - u[1]))
    C("lbu:lo", ld * f3[4] * (res[0] - sb))
    C("lbu:hi", ld * f3[4] * res[1])
    C("lhu:lo", ld * f3[5] * (res[0] - sh))
    C("lhu:hi", ld * f3[5] * res[1])
    st = opc["store"]
    nb = [sel_byte[k] * vbyte[0] + (1 - sel_byte[k]) * ubyte[k] for k in range(4)]
    C("sb:lo", st * f3[0] * (after[0] - nb[0] - 256 * nb[1]))
    C("sb:hi", st * f3[0] * (after[1] - nb[2] - 256 * nb[3]))
    C("sh:lo", st * f3[1] * (after[0] - ((1 - ob1) * v[0] + ob1 * u[0])))
    C("sh:hi", st * f3[1] * (after[1] - (ob1 * v[0] + (1 - ob1) * u[1])))
    C("sw:lo", st * f3[2] * (after[0] - v[0]))
    C("sw:hi", st * f3[2] * (after[1] - v[1]))
    # --- results of the register-writing instructions
    C("lui:lo", opc["lui"] * (res[0] - immu[0]))
    C("lui:hi", opc["lui"] * (res[1] - immu[1]))
    and_ = [lin(b, [(1 << k, ub[16 * h + k] * vb[16 * h + k]) for k in range(16)]) for h in range(2)]
    for h, nm in enumerate(("lo", "hi")):
        C("auipc:res_" + nm, opc["auipc"] * (res[h] - z[h]))
        C("link:res_" + nm, link * (res[h] - w[h]))
        C("add:res_" + nm, alu * f3[0] * (res[h] - z[h]))
        C("sll:res_" + nm, alu * f3[1] * (res[h] - z[h]))
        C("srx:res_" + nm, alu * f3[5] * (res[h] - w[h]))
        C("xor:res_" + nm, alu * f3[4] * (res[h] - (u[h] + v[h] - 2 * and_[h])))
        C("or:res_" + nm, alu * f3[6] * (res[h] - (u[h] + v[h] - and_[h])))
        C("and:res_" + nm, alu * f3[7] * (res[h] - and_[h]))
        C("mul:res_" + nm, mext * f3[0] * (res[h] - z[h]))
        C("mulh:res_" + nm, mext * (f3[1] + f3[2] + f3[3]) * (res[h] - w[h]))
        C("div:res_" + nm, isdiv * (res[h] - (bits[13] * z[h] + (1 - bits[13]) * u[h])))  # DIV[U]: the quotient (U); REM[U]: the remainder (Z)
        C("ecall:res_" + nm, opc["system"] * (res[h] - z[h]))           # what an ecall writes to a0 / a1: range-checked only
        C("ecall:word_" + nm, opc["system"] * (after[h] - w[h]))        # ... and to memory
        C("bnd:word_" + nm, bnd * (after[h] - z[h]))                    # the first value of an address is a 32-bit word
    C("slt:lo", alu * f3[2] * (res[0] - lt))
    C("slt:hi", alu * f3[2] * res[1])
    C("sltu:lo", alu * f3[3] * (res[0] - c1))
    C("sltu:hi", alu * f3[3] * res[1])
    # --- the multiplier: U (bytes) x M (byte limbs mb0..mb3) = Z + 2^32 W through four 16-bit positions; carries are range-checked
    # (access 3's digits are free on these rows: no instruction multiplies and touches memory), the sign of a signed operand is
    # folded in as -2^32 (sx M + sm U).  Shifts: M = 2^s (left) or 2^(32 - s) (right: the answer is the high word; s = 0 puts 256 in limb 3)
    p8, sx, sm, c3 = d("p8"), d("sx"), d("sm"), d("c3")
    mb = [d("mb%d" % k) for k in range(4)]
    shl, shr, mulsel = alu * f3[1], alu * f3[5], mext * (1 - bits[14])
    sgnd = 1 - bits[12]                                                 # DIV / REM are signed, DIVU / REMU are not
    inv2 = lambda k: pow(pow(2, k, P), P - 2, P)
    pow_l = (1 + vb[0]) * (1 + 3 * vb[1]) * (1 + 15 * vb[2])
    pow_r = (1 + (inv2(1) - 1) * vb[0]) * (1 + (inv2(2) - 1) * vb[1]) * (1 + (inv2(4) - 1) * vb[2])
    C("p8", p8 - (shl * pow_l + 256 * (shr * pow_r)))
    q = [(1 - vb[3]) * (1 - vb[4]), vb[3] * (1 - vb[4]), (1 - vb[3]) * vb[4], vb[3] * vb[4]]
    for j in range(4):
        C("mb%d" % j, mb[j] - (mext * vbyte[j] + p8 * (shl * q[j] + shr * q[3 - j])))
    C("sx", sx - ub[31] * (mext * (f3[1] + f3[2]) + shr * bits[30] + isdiv * sgnd))
    C("sm", sm - vb[31] * (mext * f3[1] + isdiv * sgnd))
    C("c3", c3 * (c3 + 1) * ((c3 + 2) * (c3 - 1)))
    dg = [d("d3_%d" % i) for i in range(12)] + [d("cx%d" % i) for i in range(4)]
    for i in range(4):
        digit(dg[12 + i], "cx%d" % i)
    cm = [lin(b, [(4 ** i, dg[i]) for i in range(5)]), lin(b, [(4 ** i, dg[5 + i]) for i in range(6)]), lin(b, [(4 ** i, dg[11 + i]) for i in range(5)])]
    s_ = [lin(b, [(1, ubyte[i] * mb[k - i]) for i in range(4) if 0 <= k - i < 4]) for k in range(7)]
    m_lo, m_hi = mb[0] + 256 * mb[1], mb[2] + 256 * mb[3]
    msel = mulsel + shl + shr
    C("mul:t0", msel * (s_[0] + 256 * s_[1] - z[0] - 65536 * cm[0]))
    C("mul:t1", msel * (s_[2] + 256 * s_[3] + cm[0] - z[1] - 65536 * cm[1]))
    C("mul:t2", msel * (s_[4] + 256 * s_[5] + cm[1] - sx * m_lo - sm * u[0] - w[0] - 65536 * (cm[2] - 4)))
    C("mul:t3", msel * (s_[6] + (cm[2] - 4) - sx * m_hi - sm * u[1] - w[1] - 65536 * c3))
    # --- division: U = quotient, V = divisor, Z = remainder, x[rs1] = dividend.  The same chain proves quotient x divisor + remainder =
    # dividend as 64-bit (sign-extended) integers; W = |divisor| - |remainder| - 1 is a range-checked word, so |remainder| < |divisor|;
    # a remainder other than 0 has the dividend's sign.  Division by zero: quotient all ones, the chain then gives remainder = dividend.
    # -2^31 / -1 (ovf): quotient = dividend, remainder 0.
    dv, ovf, k0, a31 = d("dv"), d("ovf"), d("k0"), d("a31")
    at = [d("at%d" % i) for i in range(8)]
    for nm, x in (("ovf", ovf), ("a31", a31)):
        bit(x, nm)
    C("k0", (k0 + 1) * k0 * ((k0 - 1) * (k0 - 2)))                     # the carry between the halves of the comparison: -1 .. 2
    for i in range(8):
        digit(at[i], "at%d" % i)
    C("dv", dv - isdiv * (1 - ovf))
    C("ovf:div", ovf * (1 - isdiv))
    C("ovf:signed", ovf * bits[12])
    C("ovf:a_lo", ovf * a[0])
    C("ovf:a_hi", ovf * (a[1] - 0x8000))
    C("ovf:b_lo", ovf * (v[0] - 0xFFFF))
    C("ovf:b_hi", ovf * (v[1] - 0xFFFF))
    for h, nm in enumerate(("lo", "hi")):
        C("ovf:q_" + nm, ovf * (u[h] - a[h]))
        C("ovf:rem_" + nm, ovf * z[h])
        C("div0:q_" + nm, dv * eq * (u[h] - 0xFFFF))
    C("div:a31", isdiv * (a[1] - 32768 * a31 - lin(b, [(4 ** i, at[i]) for i in range(8)])))
    C("div:a31_range", isdiv * at[7] * (at[7] - 1))
    C("div:rem31", isdiv * (zd[15] - 2 * c1 - c0))                      # c1 / c0: the top two bits of the remainder
    sr, sa, sb_ = c1 * sgnd, a31 * sgnd, vb[31] * sgnd
    C("div:t0", dv * (s_[0] + 256 * s_[1] + z[0] - a[0] - 65536 * cm[0]))
    C("div:t1", dv * (s_[2] + 256 * s_[3] + cm[0] + z[1] - a[1] - 65536 * cm[1]))
    C("div:t2", dv * (s_[4] + 256 * s_[5] + cm[1] - sx * m_lo - sm * u[0] + 65535 * (sr - sa) - 65536 * (cm[2] - 4)))
    C("div:t3", dv * (s_[6] + (cm[2] - 4) - sx * m_hi - sm * u[1] + 65535 * (sr - sa) - 65536 * c3))
    C("div:rem_sign", dv * (sr - sa) * (z[0] + z[1]))
    cmp = dv * (1 - eq)                                                 # ... unless the divisor is 0
    C("div:less_lo", cmp * (w[0] + 1 + (1 - 2 * sr) * z[0] - (1 - 2 * sb_) * v[0] - 65536 * k0))
    C("div:less_hi", cmp * (w[1] + (1 - 2 * sr) * z[1] - (1 - 2 * sb_) * v[1] - 65536 * (sb_ - sr) + k0))
    # --- the five accesses.  Timestamp of access k of a cycle: 5 cycle + k + 1.  An access that does not happen leaves its read
    # tuple equal to its written tuple (they cancel in the grand product); one that happens writes its own timestamp, larger than
    # the one it read: own - previous - 1 is a sum of twelve radix-4 digits.
    dig = [[d("d%d_%d" % (k, i)) for i in range(12)] for k in range(5)]
    for k in range(5):
        for i in range(12):
            digit(dig[k][i], "d%d_%d" % (k, i))

    def stamp(k):
        return 5 * cycle + (k + 1)

    def ordered(act, k):  # act * (tw_k - p_k - 1 - digits_k) = 0
        C("ordered:%d" % k, act * (d("tw%d" % k) - d("p%d" % k) - 1 - lin(b, [(4 ** i, dig[k][i]) for i in range(12)])))

    for k, (zn, invn, actn, lo_bit) in enumerate((("z1", "inv1", "act0", 15), ("z2", "inv2", "act1", 20))):
        idx = lin(b, [(1 << i, bits[lo_bit + i]) for i in range(5)])
        zk, act = d(zn), d(actn)
        C("rs%d:zero" % (k + 1), zk * idx)                              # z = 1 iff the index is 0 ...
        C("rs%d:inv" % (k + 1), idx * d(invn) - (1 - zk))
        C("rs%d:act" % (k + 1), act - live * (1 - zk) - sys_)           # x0 is not memory: no access (an ecall reads a7 / a0 here)
        for hf in ("lo", "hi"):
            C("rs%d:x0_%s" % (k + 1, hf), zk * (1 - sys_) * d("rs%d_%s" % (k + 1, hf)))  # ... and reads as zero
        C("rs%d:addr" % (k + 1), act * (d("addr%d" % k) - REG_BASE - idx - (17, 10)[k] * sys_))  # the register the word names
        C("rs%d:tw" % (k + 1), act * (d("tw%d" % k) - stamp(k)))
        C("rs%d:idle" % (k + 1), (1 - act) * (d("tw%d" % k) - d("p%d" % k)))
        ordered(act, k)
    act2, zrd = d("act2"), d("zrd")
    bit(act2, "act2")
    idx_rd = lin(b, [(1 << i, bits[7 + i]) for i in range(5)])
    C("rd:zero", zrd * idx_rd)
    C("rd:inv", idx_rd * d("inv_rd") - (1 - zrd))
    writes = opc["lui"] + opc["auipc"] + link + opc["load"] + opc["imm"] + opc["op"]
    C("rd:act", (1 - opc["system"]) * (act2 - writes * (1 - zrd)))      # an instruction with a destination other than x0 writes it
    C("rd:addr", act2 * (1 - opc["system"]) * (d("addr2") - REG_BASE - idx_rd))
    # --- ecalls: U = a7 names the function (0 HALT, 1 READ_WORDS, 2 COMMIT, 3 CYCLES, 4 PAUSE), V = a0.  The two transfers count a1
    # down: while a1 = j > 0 the cycle moves word j - 1 of the buffer at a0, writes a1 = j - 1 and repeats; with a1 = 0 it falls
    # through.  What is moved -- input words in, journal words out -- is the host's to say (as the input is in risc0): the words read in
    # are range-checked, the journal is bound by the claim's output digest outside the circuit.  CYCLES writes a0 (range-checked).
    io = d("io")
    fn_read_or_commit = ub[0] + ub[1] - 2 * ub[0] * ub[1]
    C("ecall:fn_lo", sys_ * (a[0] - ub[0] - 2 * ub[1] - 4 * ub[2]))
    C("ecall:fn_hi", sys_ * a[1])
    C("ecall:fn_max", sys_ * ub[2] * (ub[0] + ub[1]))
    C("ecall:io", io - sys_ * fn_read_or_commit * (1 - ub[2]))
    active = io * (1 - eq)                                              # eq: a1 = 0
    C("ecall:act2", sys_ * (act2 - io - ub[0] * ub[1]))                 # the transfers write a1, CYCLES writes a0, HALT / PAUSE nothing
    C("ecall:rd", act2 * sys_ * (d("addr2") - (REG_BASE + 11) + ub[0] * ub[1]))
    C("ecall:count_lo", io * (old[0] - (1 - eq) - d("new_lo") + 65536 * c0))  # a1 - 1 (a1 itself at 0) over the halves, c0 the borrow: exact in 32 bits
    C("ecall:count_hi", io * (old[1] - c0 - d("new_hi")))
    for k in (13, 14, 15):
        C("ecall:count_range_%d" % k, io * zd[k])                       # at most 2^26 words: the count is itself and not itself + p
    C("next:ecall", sys_ * (next_pc - pc - 4 + 4 * active))             # repeats while it moves, then falls through
    C("ecall:mem", sys_ * (d("mem_kind") - active * (2 * ub[0] + ub[1])))  # READ_WORDS writes memory, COMMIT reads it
    C("ecall:addr", active * (4 * d("addr3") - rs2[0] - 65536 * rs2[1] - 4 * (z[0] + 65536 * z[1])))
    C("ecall:buffer", io * (vb[0] + vb[1] + vb[30] + vb[31]))           # a0: word-aligned, below 1 GiB
    C("rd:tw", act2 * (d("tw2") - stamp(2)))
    for x_, y_ in (("tw2", "p2"), ("new_lo", "old_lo"), ("new_hi", "old_hi")):
        C("rd:idle_" + x_, (1 - act2) * (d(x_) - d(y_)))
    C("rd:lo", act2 * (d("new_lo") - res[0]))                           # ... with the result
    C("rd:hi", act2 * (d("new_hi") - res[1]))
    ordered(act2, 2)
    mk = d("mem_kind")
    C("mem:kinds", mk * (mk - 1) * (mk - 2))                            # none / read / write
    half = (P + 1) // 2
    mem_act = half * (mk * (3 - mk))                                    # 1 on reads and writes
    is_write = half * (mk * (mk - 1))                                   # 1 on writes
    keeps = 1 - is_write - bnd                                          # the word stays as it was unless written (or a boundary row)
    C("mem:keeps_lo", keeps * (after[0] - before[0]))
    C("mem:keeps_hi", keeps * (after[1] - before[1]))
    C("mem:tw", mem_act * (d("tw3") - stamp(3)))
    C("mem:idle", (1 - mem_act - bnd) * (d("tw3") - d("p3")))
    C("bnd:tw", bnd * d("tw3"))                                         # a boundary row writes the first tuple: timestamp 0
    ordered(mem_act, 3)
    # boundary rows: one history per address.  The address is below 2^28 (memory, word index) or 2^28 + a register index (digits of
    # accesses 0 / 1, free on these rows), and exceeds the previous boundary row's by 1 + fifteen digits: strictly increasing as integers
    ad = dig[0] + dig[1][:2]
    top = dig[1][2]
    C("bnd:top", bnd * top * (top - 1))
    for i in range(3, 14):
        C("bnd:reg_%d" % i, bnd * top * ad[i])
    C("bnd:addr", bnd * (d("addr3") - lin(b, [(4 ** i, ad[i]) for i in range(14)]) - (1 << 28) * top))
    gap = d("addr3") - d("addr3", 1) - 1 - lin(b, [(4 ** i, dig[3][i]) for i in range(12)] + [(4 ** (12 + i), dig[2][i]) for i in range(3)])
    C("bnd:order", bnd * prev_bnd * gap)
    C("fetch:addr", live * (pc - 4 * d("addr4")))                       # the fetch reads the word at pc
    C("fetch:tw", live * (d("tw4") - stamp(4)))
    C("fetch:idle", not_live * (d("tw4") - d("p4")))
    ordered(live, 4)
    # --- public inputs: the run starts at pc0 in cycle 0; the row after the last cycle (or the last row itself) pins the end
    G0 = 8
    gl = lambda k: E(b, b.glob(0, k), 0)
    C("first:live", first * (live - 1))
    C("first:pc", first * (pc - gl(G0)))
    C("first:cycle", first * cycle)
    ended = not_first * (prev_live - live)                              # 1 on the first row that is not a cycle
    C("end:pc", ended * (d("next_pc", 1) - gl(G0 + 1)))
    C("end:cycles", ended * (d("cycle", 1) + 1 - gl(G0 + 2)))
    full = last * live                                                  # a trace that fills every row
    C("full:pc", full * (next_pc - gl(G0 + 1)))
    C("full:cycles", full * (cycle + 1 - gl(G0 + 2)))
    # --- how the segment ends: a HALT / PAUSE ecall is the last cycle of its segment, and the public inputs say which it was (0: the
    # segment was cut, 1: HALT, 2: PAUSE; G0 + 4 is "not 0", checked against it by the verifier) and with which exit code (a0)
    ub_p = [d("ub%d" % k, 1) for k in range(3)]
    term_prev = d("opc_system", 1) * (1 - ub_p[0]) * (1 - ub_p[1])      # the row before was a HALT (a7 = 0) or a PAUSE (a7 = 4)
    term_here = sys_ * (1 - ub[0]) * (1 - ub[1])
    C("exit:last_cycle", not_first * term_prev * live)
    for tag, gate, term, u2, lo, hi in (("end", ended, term_prev, ub_p[2], d("rs2_lo", 1), d("rs2_hi", 1)), ("full", full, term_here, ub[2], rs2[0], rs2[1])):
        C("exit:%s_is" % tag, gate * (term - gl(G0 + 4)))
        C("exit:%s_kind" % tag, gate * gl(G0 + 4) * (1 + u2 - gl(G0 + 3)))
        C("exit:%s_none" % tag, gate * (1 - gl(G0 + 4)) * gl(G0 + 3))
        C("exit:%s_lo" % tag, gate * (gl(G0 + 4) * lo - gl(G0 + 5)))
        C("exit:%s_hi" % tag, gate * (gl(G0 + 4) * hi - gl(G0 + 6)))
    # --- the grand products: RS_A, RS_B over the tuples read, WS_A, WS_B over the tuples written
    alpha = [b.glob(1, i) for i in range(4)]
    beta = [[b.glob(1, 4 * (j + 1) + i) for i in range(4)] for j in range(3)]
    dv = lambda name, back=0: b.get(G_DATA, col[name], back)

    def fingerprint(addr, lo, hi, t):  # alpha - addr - b1 lo - b2 hi - b3 t as four base-field expressions
        out = []
        for i in range(4):
            e = b.sub(alpha[i], b.add(b.add(b.mul(beta[0][i], dv(lo)), b.mul(beta[1][i], dv(hi))), b.mul(beta[2][i], dv(t))))
            out.append(b.sub(e, dv(addr)) if i == 0 else e)
        return out

    products = [  # (ACCUM column block, the fingerprints it multiplies) -- mirrored in the SEC_ACCUM_FP records
        [("addr0", "rs1_lo", "rs1_hi", "p0"), ("addr1", "rs2_lo", "rs2_hi", "p1"), ("addr2", "old_lo", "old_hi", "p2")],
        [("addr3", "before_lo", "before_hi", "p3"), ("addr4", "insn_lo", "insn_hi", "p4")],
        [("addr0", "rs1_lo", "rs1_hi", "tw0"), ("addr1", "rs2_lo", "rs2_hi", "tw1"), ("addr2", "new_lo", "new_hi", "tw2")],
        [("addr3", "after_lo", "after_hi", "tw3"), ("addr4", "insn_lo", "insn_hi", "tw4")],
    ]
    one = b.const(1)
    nf = b.sub(one, first.v)
    for j, tuples in enumerate(products):
        prev = [b.get(G_ACCUM, 4 * j + i, 1) for i in range(4)]
        want = [b.mul(nf, prev[i]) for i in range(4)]
        want[0] = b.add(want[0], first.v)
        for t in tuples:
            want = fp4_mul_sym(b, want, fingerprint(*t))
        for i in range(4):
            cons.append(("accum:%d_%d" % (j, i), b.sub(b.get(G_ACCUM, 4 * j + i, 0), want[i]), 2 + len(tuples), True))
    acc = [[b.get(G_ACCUM, 4 * j + i, 0) for i in range(4)] for j in range(4)]
    reads, writes_ = fp4_mul_sym(b, acc[0], acc[1]), fp4_mul_sym(b, acc[2], acc[3])
    for i in range(4):
        cons.append(("accum:equal_%d" % i, b.mul(last.v, b.sub(reads[i], writes_[i])), 3, True))  # every tuple read was written, once
    assert max(c[2] for c in cons) <= 5
    return b, cons, products


def generate_trace():
    col = {name: i for i, name in enumerate(TRACE_COLUMNS)}
    n_data, n_code, n_acc, n_global = len(TRACE_COLUMNS), 4, 4, TRACE_GLOBALS
    code_cols = [(0, 0), (1, 0), (2, 0), (3, 3)]  # first-row indicator, last-row indicator, row index, one seeded column
    data_cols = [(0, 0, 0, 0, 0)] * n_data        # all free: the witness is the execution's
    b, cons, products = trace_constraints()
    x = b.true()
    for _, v, _, _ in cons:
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
    words += section(SEC_INFO, list(struct.unpack("<4I", b"R0HIP_TRACE:v3__")))
    words += section(SEC_GROUPS, [4 * n_acc, n_code, n_data])
    words += section(SEC_TAPS, [len(taps)] + [w for t in taps for w in t])
    words += section(SEC_GLOBALS, [n_global, 16])
    words += section(SEC_POLY, [len(steps), x] + [w for st in steps for w in st])
    words += section(SEC_WITGEN, [n_code] + [w for cc in code_cols for w in cc] + [n_data] + [w for dc in data_cols for w in dc])
    words += section(SEC_ACCUM_FP, [n_acc] + acc_records)
    info = {"taps": len(taps), "steps": len(steps), "constraints": len(cons), "mul_per_point": b.n_mul, "addsub_per_point": b.n_add,
            "groups": [4 * n_acc, n_code, n_data], "columns": len(TRACE_COLUMNS)}
    return words, info


def check_trace_rows(data, globals_, first_only=True):
    """Evaluate every DATA / CODE constraint of the trace circuit on a witness: data[column][row] canonical integers (numpy int64),
    globals_ the TRACE_GLOBALS public inputs as canonical integers.  -> [(constraint name, rows where it does not vanish)].  A development
    and test aid: tells WHICH constraint a witness breaks, where the prover only says that one does."""
    import numpy as np
    b, cons, _ = trace_constraints()
    n = data.shape[1]
    rows = np.arange(n)
    code = [(rows == 0).astype(np.int64), (rows == n - 1).astype(np.int64), rows.astype(np.int64), np.zeros(n, dtype=np.int64)]
    wanted = {v for _, v, _, accum in cons if not accum}
    vals, fp = {}, 0
    zero = np.zeros(n, dtype=np.int64)
    for op, a_, b_, c_ in b.steps:
        if op in (OP_TRUE, OP_AND_EQZ, OP_AND_COND):
            continue
        if op == OP_CONST:
            r = np.full(n, a_, dtype=np.int64)
        elif op == OP_GET:
            _, g, c, back = a_
            src = data[c] if g == G_DATA else code[c] if g == G_CODE else zero
            r = np.roll(src, back)
        elif op == OP_GET_GLOBAL:
            r = np.full(n, int(globals_[b_]) if a_ == 0 else 0, dtype=np.int64)
        elif op == OP_ADD:
            r = (vals[a_] + vals[b_]) % P
        elif op == OP_SUB:
            r = (vals[a_] - vals[b_]) % P
        else:
            r = (vals[a_] * vals[b_]) % P
        vals[fp] = r
        fp += 1
    bad = []
    for name, v, _, accum in cons:
        if accum:
            continue
        where = np.nonzero(vals[v])[0]
        if len(where):
            bad.append((name, where[:8].tolist() if first_only else where.tolist()))
    return bad


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
