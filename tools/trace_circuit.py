#!/usr/bin/env python3
"""The trace circuit, version 5 (ProtocolInfo "R0HIP_TRACE:v5__"; v4 with ten columns expressed as linear forms of the others: 128 DATA
columns): the circuit whose DATA group is the executor's preflight trace
(include/r0hip.h: r0h_preflight_row / r0h_preflight_bound; csrc/trace.hpp holds the same column list and fills it).

It stands where risc0's rv32im circuit stands (risc0-circuit-rv32im 4.0.4, Cargo.lock:3087-3089); it is NOT
that circuit -- its tap table and polynomial are machine-generated and not reproducible here -- but one written for this library's
executor.  What changed against v3:

  * LOOKUPS (log-derivative argument): range checks and byte logic go through two 2^16-row tables in the CODE
    group -- R16 (the values 0..65535) and AND8 (a + 256 b + 65536 (a & b), tagged with 2^24) -- instead of bit and radix-4 digit
    columns: timestamps differences are one 16-bit and one 8-bit limb, Z / W are 16-bit halves, U / V are bytes.  288 -> 138 DATA
    columns.  Every fraction multiplicity / (alpha2 - value) is one slot of a running sum in ACCUM; four slots per accumulator
    (degree 5), the accumulators of a row are chained into ONE running sum that wraps around the trace, so its total is zero.
  * MEMORY as fractions of the same sum: every access consumes the tuple (address, value, timestamp, space) its predecessor
    produced; the fetch is the EARLIEST access of a cycle (a load of the instruction's own word is consistent); x0 is an ordinary
    register whose value nobody can change.
  * THE SESSION-WIDE ARGUMENT: a boundary row consumes (address, first value, previous segment) and produces
    (address, last value, this segment) under a challenge gamma derived from the DATA roots of ALL segments of the session (late
    public inputs, absorbed after the DATA commitment); the rows of a closing segment produce the initial tuple (address, initial
    value, 0) instead, in strictly increasing address order, and name the words that belong to the program image; an active
    COMMIT row names the journal word it reads.  The segment's sum G_k is a public input: the verifier adds the image and the
    journal itself and checks that the session balances.
"""
import struct

P = 15 * 2**27 + 1
G_ACCUM, G_CODE, G_DATA = 0, 1, 2
REG_BASE = 1 << 28
TAG_AND = 1 << 24
TAG_IMG, TAG_JRN = (1 << 20) + 1, (1 << 20) + 2   # the "segment" coordinate of an image / journal tuple (segment indices are 16-bit)
SEC_LATE, SEC_LOGUP = 9, 10
TABLE_R16, TABLE_AND = 1, 2
MIN_PO2 = 16

OPCODES = [("lui", 0x37), ("auipc", 0x17), ("jal", 0x6F), ("jalr", 0x67), ("branch", 0x63), ("load", 0x03), ("store", 0x23), ("imm", 0x13),
           ("op", 0x33), ("system", 0x73), ("fence", 0x0F)]
# Ten quantities the constraints speak of are LINEAR FORMS of other columns, not columns (DERIVED below): the last flag of each one-hot
# group, V's low byte (the sum of its bits), Z's low half (its two low bits and the rest), mem_act and io (sums of the flags that
# imply them), and the timestamps accesses 0, 1, 2 and 4 consume (own timestamp - 1 - the two looked-up limbs of the difference).
# That is what brings the DATA group to 128 columns: eight Poseidon2 permutations per row of the commitment instead of nine.
TRACE_COLUMNS = (["live", "bnd", "cycle", "pc", "next_pc"]
                 + ["opc_" + name for name, _ in OPCODES[:-1]]                                 # one-hot opcode (FENCE: live minus the others)
                 + ["f3_%d" % k for k in range(1, 8)]                                          # one-hot funct3 (funct3 = 0: one minus the others)
                 + ["alu"]                                                                     # OP-IMM or base-ISA OP
                 + ["rd0", "rdA", "rdB", "r10", "r1A", "r1B", "r20", "r2A", "r2B"]           # rd / rs1 / rs2: low bit and two radix-4 digits
                 + ["b25", "f7A", "f7B", "b30", "b31"]                                         # funct7: bit 25, bits 26..29 as two digits, bits 30, 31
                 + ["rs1_lo", "rs1_hi", "dl0", "dh0"]                                          # access 0: x[rs1] (an ecall: a7)
                 + ["rs2_lo", "rs2_hi", "dl1", "dh1"]                                          # access 1: x[rs2] (an ecall: a0)
                 + ["zrd", "inv_rd", "act2", "old_lo", "old_hi", "dl2", "dh2"]                 # access 2: x[rd] written with res
                 + ["mem_wr", "top", "addr3", "before_lo", "before_hi", "after_lo", "after_hi", "p3", "dl3", "dh3"]  # access 3: the memory word / a boundary row
                 + ["dl4", "dh4"]                                                              # access 4: the fetch
                 + ["u%d" % k for k in range(4)] + ["v%d" % k for k in range(1, 4)] + ["a%d" % k for k in range(4)]  # U, V and U & V, byte by byte (V's low byte: its bits)
                 + ["su", "sv"]                                                                # their sign bits
                 + ["sh%d" % k for k in range(5)] + ["vrd", "vrb"]                             # V's low byte: five bits, a digit, a bit
                 + ["z_hi", "ob0", "ob1", "zq", "w_lo", "w_hi", "aux0", "aux1"]                # words Z and W as halves; Z's two low bits and the rest of its low half; two spare 16-bit range checks
                 + ["res_lo", "res_hi", "c0", "c1", "lt", "eq", "zinv", "sb", "sgn", "p8", "sx", "sm"]
                 + ["mb%d" % k for k in range(4)]                                              # second multiplier operand, byte limbs
                 + ["ce0", "ce1a", "ce1b", "ce2", "cb1", "cb2", "cband", "c3"]                 # the multiplier's carries (carry 0's low byte sits in dh3)
                 + ["dv", "ovf", "k0", "a31"]                                                  # division
                 + ["f0", "f1", "f2", "fn_cyc", "cact"]                                        # ecalls: function bits; CYCLES; an active COMMIT
                 + ["fimg"]                                                                    # a closing boundary row whose word belongs to the program image
                 + ["m16", "mand"])                                                            # multiplicities of the two tables
assert len(TRACE_COLUMNS) == 128
COL = {name: i for i, name in enumerate(TRACE_COLUMNS)}
N_CODE = 6            # first row, last row, row index, one seeded column, R16, AND8
CODE_T16, CODE_TAND = 4, 5
# public inputs: claim words 0..7, first pc, pc after the last cycle, cycles, how the segment ends (0 cut / 1 HALT / 2 PAUSE), that being
# non-zero, exit code halves, this segment's number k >= 1, closing flag, k (1 - closing), first / last boundary address; then LATE:
# alpha_g, gamma1..3 (16 words), the segment's sum G_k (4 words)
G_PC0, G_PC1, G_CYCLES, G_KIND, G_TERM, G_EXIT_LO, G_EXIT_HI, G_SEG, G_FIN, G_SEGNF, G_ALO, G_AHI = range(8, 20)
G_GAMMA, G_SUM = 20, 36
TRACE_GLOBALS, TRACE_LATE = 40, 20
# the mix drawn after the late globals: alpha, beta1..4 (memory tuples), alpha2 (lookups)
MIX_ALPHA, MIX_B1, MIX_B2, MIX_B3, MIX_B4, MIX_ALPHA2 = range(6)
TRACE_MIX = 24
STAMP = {"fetch": 1, "rs1": 2, "rs2": 3, "rd": 4, "mem": 5}  # timestamp of an access of cycle c: 5 c + STAMP


class LF:
    """A base-field linear form: sum of coef * [global] * [column]; keys are (global index or None, column ref or None)."""
    __slots__ = ("t",)

    def __init__(self, t=None):
        self.t = {k: v % P for k, v in (t or {}).items() if v % P}

    @staticmethod
    def of(x):
        return x if isinstance(x, LF) else LF({(None, None): x})

    @staticmethod
    def col(name, group=G_DATA):
        return LF({(None, (group << 28) | (COL[name] if isinstance(name, str) else name)): 1})

    @staticmethod
    def glob(idx):
        return LF({(idx, None): 1})

    def __add__(self, o):
        o = LF.of(o)
        t = dict(self.t)
        for k, v in o.t.items():
            t[k] = (t.get(k, 0) + v) % P
        return LF(t)

    __radd__ = __add__

    def __neg__(self):
        return LF({k: -v for k, v in self.t.items()})

    def __sub__(self, o):
        return self + (-LF.of(o))

    def __rsub__(self, o):
        return LF.of(o) - self

    def __mul__(self, o):
        if isinstance(o, LF):  # only (pure global) x (anything without a global), or a constant
            a, b = (self, o) if all(k[1] is None for k in self.t) else (o, self)
            assert all(k[1] is None for k in a.t), "product of two column forms is not linear"
            t = {}
            for (ga, _), va in a.t.items():
                for (gb, cb), vb in b.t.items():
                    assert ga is None or gb is None, "product of two globals"
                    key = (ga if ga is not None else gb, cb)
                    t[key] = (t.get(key, 0) + va * vb) % P
            return LF(t)
        return LF({k: v * o for k, v in self.t.items()})

    __rmul__ = __mul__

    def words(self):
        out = [len(self.t)]
        for (g, c), v in sorted(self.t.items(), key=lambda kv: (kv[0][0] is not None, kv[0][0] or 0, kv[0][1] is not None, kv[0][1] or 0)):
            out += [v, 0 if g is None else g + 1, 0 if c is None else c + 1]
        return out

    def expr(self, b, E):
        """-> an E (constraint expression) over Builder b"""
        acc = E.of(b, 0)
        for (g, c), v in sorted(self.t.items(), key=lambda kv: (kv[0][0] is not None, kv[0][0] or 0, kv[0][1] is not None, kv[0][1] or 0)):
            term = E.of(b, v)
            if g is not None:
                term = term * E(b, b.glob(0, g), 0)
            if c is not None:
                term = term * E(b, b.get(c >> 28, c & 0xFFFFF, 0), 1)
            acc = acc + term
        return acc

    def evaluate(self, data, code, globals_):
        """numpy: the form on every row (canonical integers mod p, int64)"""
        import numpy as np
        n = data.shape[1]
        acc = np.zeros(n, dtype=np.int64)
        for (g, c), v in self.t.items():
            k = v * (int(globals_[g]) if g is not None else 1) % P
            if c is None:
                acc = (acc + k) % P
            else:
                src = data[c & 0xFFFFF] if (c >> 28) == G_DATA else code[c & 0xFFFFF]
                acc = (acc + k * (src.astype(np.int64) % P)) % P  # both factors below 2^31
        return acc


ONE = LF.of(1)


class Fraction:
    """numerator / (sum over parts of challenge x form).  A part's challenge is ('one',), ('mix', i) or ('glob', first of 4 globals)."""

    def __init__(self, name, num, parts, table=0):
        self.name, self.num, self.parts, self.table = name, LF.of(num), [(ch, LF.of(f)) for ch, f in parts], table

    def words(self):
        out = [self.table] + self.num.words() + [len(self.parts)]
        for ch, f in self.parts:
            out += [{"one": 0, "mix": 1, "glob": 2}[ch[0]], ch[1] if len(ch) > 1 else 0] + f.words()
        return out


_DERIVED = {}


def derived():
    """name -> linear form, for the quantities that are not columns of their own"""
    if not _DERIVED:
        col = LF.col
        d = _DERIVED
        d["opc_fence"] = col("live") - sum((col("opc_" + name) for name, _ in OPCODES[:-1]), LF())
        d["f3_0"] = 1 - sum((col("f3_%d" % k) for k in range(1, 8)), LF())
        d["v0"] = sum(((1 << k) * col("sh%d" % k) for k in range(5)), LF()) + 32 * col("vrd") + 128 * col("vrb")
        d["z_lo"] = col("ob0") + 2 * col("ob1") + 4 * col("zq")
        d["mem_act"] = col("opc_load") + col("mem_wr") + col("cact")   # a load, a store or READ_WORDS moving a word (mem_wr), a COMMIT moving one
        d["io"] = col("f0") + col("f1") - 2 * col("fn_cyc")            # a7 = 1 or 2 (fn_cyc = f0 f1; f2 excludes both)
        for k, key in ((0, "rs1"), (1, "rs2"), (2, "rd"), (4, "fetch")):  # the timestamp an access consumes: smaller than its own by 1 + the two looked-up limbs
            d["p%d" % k] = 5 * col("cycle") + STAMP[key] - 1 - col("dl%d" % k) - 65536 * col("dh%d" % k)
    return _DERIVED


def c(name):
    return derived()[name] if name in derived() else LF.col(name)


def mem_tuple(name, num, addr, lo, hi, ts, space):
    """a memory tuple's fraction: num / (alpha - addr - b1 lo - b2 hi - b3 ts - b4 space)"""
    return Fraction(name, num, [(("mix", MIX_ALPHA), ONE), (("one",), -LF.of(addr)), (("mix", MIX_B1), -LF.of(lo)), (("mix", MIX_B2), -LF.of(hi)),
                                (("mix", MIX_B3), -LF.of(ts)), (("mix", MIX_B4), -LF.of(space))])


def session_tuple(name, num, addr, lo, hi, t):
    """a session-wide tuple's fraction under the late public challenge: num / (alpha_g - addr - g1 lo - g2 hi - g3 t)"""
    return Fraction(name, num, [(("glob", G_GAMMA), ONE), (("one",), -LF.of(addr)), (("glob", G_GAMMA + 4), -LF.of(lo)), (("glob", G_GAMMA + 8), -LF.of(hi)),
                                (("glob", G_GAMMA + 12), -LF.of(t))])


def lookup16(name, value, num=1):
    return Fraction(name, num, [(("mix", MIX_ALPHA2), ONE), (("one",), -LF.of(value))], TABLE_R16)


def lookup_and(name, a, b, r, num=1):
    return Fraction(name, num, [(("mix", MIX_ALPHA2), ONE), (("one",), -(LF.of(a) + 256 * LF.of(b) + 65536 * LF.of(r) + TAG_AND))], TABLE_AND)


def decode_forms():
    """linear forms of the instruction word's fields over the decode columns"""
    f = {}
    f["rd"] = c("rd0") + 2 * c("rdA") + 8 * c("rdB")
    f["rs1"] = c("r10") + 2 * c("r1A") + 8 * c("r1B")
    f["rs2"] = c("r20") + 2 * c("r2A") + 8 * c("r2B")
    f["f7lo"] = c("b25") + 2 * c("f7A") + 8 * c("f7B") + 32 * c("b30")          # instruction bits 25..30
    f["f3v"] = sum((k * c("f3_%d" % k) for k in range(1, 8)), LF())
    f["opc7"] = sum((code * c("opc_" + name) for name, code in OPCODES), LF())
    f["insn_lo"] = f["opc7"] + 128 * f["rd"] + 4096 * f["f3v"] + 32768 * c("r10")
    f["insn_hi"] = c("r1A") + 4 * c("r1B") + 16 * f["rs2"] + 512 * (f["f7lo"] + 64 * c("b31"))
    f["bit12"] = c("f3_1") + c("f3_3") + c("f3_5") + c("f3_7")
    f["bit13"] = c("f3_2") + c("f3_3") + c("f3_6") + c("f3_7")
    f["bit14"] = c("f3_4") + c("f3_5") + c("f3_6") + c("f3_7")
    return f


def fractions():
    """-> (chained accumulators' fractions in slot order, the session accumulator's fractions): the whole log-derivative argument"""
    d = decode_forms()
    live, bnd, cyc, sys_ = c("live"), c("bnd"), c("cycle"), c("opc_system")
    st = lambda k: 5 * cyc + STAMP[k]
    reg = lambda idx: REG_BASE + idx
    a0, a1 = reg(d["rs1"] + 17 * sys_), reg(d["rs2"] + 10 * sys_)
    a2 = reg(d["rd"] + 11 * sys_ - c("fn_cyc"))
    inv4 = pow(4, P - 2, P)
    a4 = inv4 * c("pc")
    memnum = c("mem_act") + bnd
    mem = [
        mem_tuple("rs1:read", live, a0, c("rs1_lo"), c("rs1_hi"), c("p0"), 1),
        mem_tuple("rs1:write", -live, a0, c("rs1_lo"), c("rs1_hi"), st("rs1") * 1 - 0, 1),
        mem_tuple("rs2:read", live, a1, c("rs2_lo"), c("rs2_hi"), c("p1"), 1),
        mem_tuple("rs2:write", -live, a1, c("rs2_lo"), c("rs2_hi"), st("rs2"), 1),
        mem_tuple("rd:read", c("act2"), a2, c("old_lo"), c("old_hi"), c("p2"), 1),
        mem_tuple("rd:write", -c("act2"), a2, c("res_lo"), c("res_hi"), st("rd"), 1),
        mem_tuple("mem:read", memnum, c("addr3"), c("before_lo"), c("before_hi"), c("p3"), c("top")),           # a boundary row reads its address's last tuple
        mem_tuple("mem:write", -memnum, c("addr3"), c("after_lo"), c("after_hi"), 5 * cyc + 5 * c("mem_act"), c("top")),  # ... and writes its first, timestamp 0
        mem_tuple("fetch:read", live, a4, d["insn_lo"], d["insn_hi"], c("p4"), 0),
        mem_tuple("fetch:write", -live, a4, d["insn_lo"], d["insn_hi"], st("fetch"), 0),
    ]
    look = [lookup16("dl%d" % k, c("dl%d" % k)) for k in range(5)]
    look += [lookup_and("dh%d" % k, c("dh%d" % k), 0, 0) for k in range(5)]
    look += [lookup_and("and%d" % k, c("u%d" % k), c("v%d" % k), c("a%d" % k)) for k in range(4)]
    u_hi, v_hi = c("u2") + 256 * c("u3"), c("v2") + 256 * c("v3")
    look += [lookup16("su", 2 * (u_hi - 32768 * c("su"))), lookup16("sv", 2 * (v_hi - 32768 * c("sv")))]
    look += [lookup16(n, c(n)) for n in ("z_lo", "z_hi", "zq", "w_lo", "w_hi", "aux0", "aux1")]
    look += [lookup_and("carry", c("cb1"), c("cb2"), c("cband"))]
    look += [Fraction("table:r16", -c("m16"), [(("mix", MIX_ALPHA2), ONE), (("one",), -LF.col(CODE_T16, G_CODE))]),
             Fraction("table:and", -c("mand"), [(("mix", MIX_ALPHA2), ONE), (("one",), -LF.col(CODE_TAND, G_CODE))])]
    fin, k_nf = LF.glob(G_FIN), LF.glob(G_SEGNF)
    pv = lambda h: c("before_" + h) - fin * c("before_" + h) + fin * c("old_" + h)  # what a boundary row hands on: the last value, or -- closing -- the initial one
    session = [
        session_tuple("session:consume", bnd, c("addr3"), c("after_lo"), c("after_hi"), LF.glob(G_SEG) - 1 - c("dl2")),  # (address, first value, the segment that left it: an EARLIER one, by a looked-up gap)
        session_tuple("session:produce", -bnd, c("addr3"), pv("lo"), pv("hi"), k_nf),                      # (address, last value, this segment) / (address, initial value, 0)
        session_tuple("session:image", c("fimg"), c("addr3"), c("old_lo"), c("old_hi"), TAG_IMG),           # an image word's initial value: the verifier holds the other side
        session_tuple("session:journal", c("cact"), c("addr3"), c("before_lo"), c("before_hi"), TAG_JRN),   # a journal word: likewise
    ]
    return mem + look, session


def accumulators():
    """-> [(fractions of the accumulator (<= 4), final: None = link of the chain / first of the 4 globals its total is)]"""
    chained, session = fractions()
    assert len(chained) % 4 == 0, len(chained)
    accs = [(chained[i:i + 4], None) for i in range(0, len(chained), 4)]
    accs.append((session, G_SUM))
    return accs


N_ACC = 10


def logup_section():
    accs = accumulators()
    assert len(accs) == N_ACC
    words = [len(accs), 2, COL["m16"], TABLE_R16, COL["mand"], TABLE_AND]
    for fr, final in accs:
        words += [len(fr), 0xFFFFFFFF if final is None else final]
        for f in fr:
            words += f.words()
    return words


def accum_constraints(b, E, fp4_mul_sym, accs, first, cons):
    """The running sums of a log-derivative argument as constraints (appended to `cons`): accumulator j of a row adds its four
    fractions to accumulator j - 1 of the same row, the first one to the last one of the row before -- around the end of the trace as
    well, so the chain's total is zero; an accumulator with a public total runs on its own and wraps with that total.
    accs: [(four Fractions, None | index of the first of the 4 public inputs its total is)]; first: the first-row indicator (an E)."""
    n_chain = sum(1 for _, fin in accs if fin is None)

    def ch_vars(ch):
        if ch[0] == "mix":
            return [b.glob(1, 4 * ch[1] + i) for i in range(4)]
        if ch[0] == "glob":
            return [b.glob(0, ch[1] + i) for i in range(4)]
        return None

    def fp4_scale(x, e):  # Fp4 (list of 4 fp vars) times a base expression
        return list(x) if e.k == 1 else [b.mul(xi, e.v) for xi in x]

    def fp4_add(x, y):
        return [b.add(xi, yi) for xi, yi in zip(x, y)]

    for j, (fr, final) in enumerate(accs):
        dens, nums = [], []
        for f in fr:
            den = None
            for ch, lf in f.parts:
                e = lf.expr(b, E)
                cv = ch_vars(ch)
                if cv is None:
                    part = [e.v, None, None, None]
                else:
                    part = [b.mul(x, e.v) if e.k != 1 else x for x in cv]
                if den is None:
                    den = [p_ if p_ is not None else b.const(0) for p_ in part]
                else:
                    den = [b.add(dq, pq) if pq is not None else dq for dq, pq in zip(den, part)]
            dens.append(den)
            nums.append(f.num.expr(b, E))
        assert len(dens) == 4
        d01, d23 = fp4_mul_sym(b, dens[0], dens[1]), fp4_mul_sym(b, dens[2], dens[3])
        big = fp4_mul_sym(b, d01, d23)
        n01 = fp4_add(fp4_scale(dens[1], nums[0]), fp4_scale(dens[0], nums[1]))   # n0 d1 + n1 d0
        n23 = fp4_add(fp4_scale(dens[3], nums[2]), fp4_scale(dens[2], nums[3]))
        numer = fp4_add(fp4_mul_sym(b, n01, d23), fp4_mul_sym(b, n23, d01))
        cur = [b.get(G_ACCUM, 4 * j + i, 0) for i in range(4)]
        if final is None:
            prev = [b.get(G_ACCUM, 4 * (n_chain - 1) + i, 1) for i in range(4)] if j == 0 else [b.get(G_ACCUM, 4 * (j - 1) + i, 0) for i in range(4)]
            diff = [b.sub(cur[i], prev[i]) for i in range(4)]
        else:
            prev = [b.get(G_ACCUM, 4 * j + i, 1) for i in range(4)]
            diff = [b.add(b.sub(cur[i], prev[i]), b.mul(first.v, b.glob(0, final + i))) for i in range(4)]
        lhs = fp4_mul_sym(b, diff, big)
        for i in range(4):
            cons.append(("accum:%d_%d" % (j, i), b.sub(lhs[i], numer[i]), 5, True))


def trace_constraints(builder_cls, E, lin, fp4_mul_sym):
    """-> (Builder, [(name, fp var, degree, touches ACCUM)]): every polynomial that must vanish on every row of a trace"""
    b = builder_cls()
    n_data = len(TRACE_COLUMNS)
    for g, size in ((G_ACCUM, 4 * N_ACC), (G_CODE, N_CODE), (G_DATA, n_data)):
        for cc in range(size):
            b.taps.add((g, cc, 0))
    cons = []

    memo = {}

    def d(name, back=0):
        if name in derived():
            assert back == 0, name
            if name not in memo:
                memo[name] = derived()[name].expr(b, E)
            return memo[name]
        return E(b, b.get(G_DATA, COL[name], back), 1)

    def C(name, e, accum=False):
        assert e.deg <= 5, (name, e.deg)
        cons.append((name, e.v, e.deg, accum))

    def bit(v, name):
        C("bit:" + name, v * (v - 1))

    def digit(v, name):
        C("digit:" + name, v * (v - 1) * ((v - 2) * (v - 3)))

    def form(f):
        return f.expr(b, E)

    gl = lambda k: E(b, b.glob(0, k), 0)
    first, last = E(b, b.get(G_CODE, 0, 0), 1), E(b, b.get(G_CODE, 1, 0), 1)
    not_first = 1 - first
    live, prev_live, bnd, prev_bnd = d("live"), d("live", 1), d("bnd"), d("bnd", 1)
    not_live = 1 - live
    pc, next_pc, cycle = d("pc"), d("next_pc"), d("cycle")
    # --- the run: live rows first, then boundary rows, then blank rows
    bit(live, "live")
    bit(bnd, "bnd")
    C("live.bnd", live * bnd)
    gate = not_first * live
    C("run:pc", gate * (pc - d("next_pc", 1)))
    C("run:cycle", gate * (cycle - d("cycle", 1) - 1))
    C("run:after_live", gate * (1 - prev_live))
    C("run:bnd_after", not_first * bnd * (1 - prev_live - prev_bnd))
    for name in ("pc", "next_pc", "cycle", "mem_act", "act2"):
        C("idle:" + name, not_live * d(name))
    # --- decoding: fields instead of bits; the word itself is a linear form (the fetch compares it with memory)
    df = {k: form(v) for k, v in decode_forms().items()}
    for name in ("rd0", "r10", "r20", "b25", "b30", "b31"):
        bit(d(name), name)
    for name in ("rdA", "rdB", "r1A", "r1B", "r2A", "r2B", "f7A", "f7B"):
        digit(d(name), name)
    opc = {name: d("opc_" + name) for name, _ in OPCODES}
    for name, v in opc.items():
        bit(v, "opc_" + name)
    f3 = [d("f3_%d" % k) for k in range(8)]
    for k, v in enumerate(f3):
        bit(v, "f3_%d" % k)
    b25, b30, b31, f7A, f7B = d("b25"), d("b30"), d("b31"), d("f7A"), d("f7B")
    bit12, bit13, bit14 = df["bit12"], df["bit13"], df["bit14"]
    alu, mext = d("alu"), opc["op"] * b25
    C("alu", alu - (opc["imm"] + opc["op"] * (1 - b25)))
    for nm, x in (("26_27", f7A), ("28_29", f7B), ("31", b31)):
        C("op:f7_" + nm, opc["op"] * x)
    C("op:f7_m_alt", opc["op"] * b25 * b30)
    C("op:f7_alt", opc["op"] * b30 * (1 - f3[0] - f3[5]))
    for nm, x in (("25", b25), ("26_27", f7A), ("28_29", f7B), ("30", b30), ("31", b31)):
        C("slli:f7_" + nm, opc["imm"] * f3[1] * x)
        if nm != "30":
            C("srxi:f7_" + nm, opc["imm"] * f3[5] * x)
    C("jalr:f3", opc["jalr"] * (1 - f3[0]))
    C("branch:f3", opc["branch"] * (f3[2] + f3[3]))
    C("load:f3", opc["load"] * (f3[3] + f3[6] + f3[7]))
    C("store:f3", opc["store"] * (1 - f3[0] - f3[1] - f3[2]))
    C("system:lo", opc["system"] * (df["insn_lo"] - 0x73))             # ecall is the one SYSTEM word that runs
    C("system:hi", opc["system"] * df["insn_hi"])
    sign = b31
    immi = [df["rs2"] + 32 * df["f7lo"] + 0xF800 * sign, 0xFFFF * sign]     # sign-extended, as two halves
    imms = [df["rd"] + 32 * df["f7lo"] + 0xF800 * sign, 0xFFFF * sign]
    immu = [4096 * df["f3v"] + 32768 * d("r10"), df["insn_hi"]]   # instruction bits 12..31
    imm_j = 4096 * (df["f3v"] + 8 * df["rs1"]) + 2048 * d("r20") + 2 * (d("r2A") + 4 * d("r2B") + 16 * df["f7lo"]) - (1 << 20) * sign
    imm_b = 2048 * d("rd0") + 32 * df["f7lo"] + 2 * (d("rdA") + 4 * d("rdB")) - (1 << 12) * sign
    # --- the words the units work on
    ub, vb, ab = [d("u%d" % k) for k in range(4)], [d("v%d" % k) for k in range(4)], [d("a%d" % k) for k in range(4)]
    u = [ub[0] + 256 * ub[1], ub[2] + 256 * ub[3]]
    v = [vb[0] + 256 * vb[1], vb[2] + 256 * vb[3]]
    and_ = [ab[0] + 256 * ab[1], ab[2] + 256 * ab[3]]
    su, sv = d("su"), d("sv")
    bit(su, "su")
    bit(sv, "sv")
    sh = [d("sh%d" % k) for k in range(5)]
    for k in range(5):
        bit(sh[k], "sh%d" % k)
    digit(d("vrd"), "vrd")
    bit(d("vrb"), "vrb")
    z, w = [d("z_lo"), d("z_hi")], [d("w_lo"), d("w_hi")]
    a = [d("rs1_lo"), d("rs1_hi")]
    rs2 = [d("rs2_lo"), d("rs2_hi")]
    before, after = [d("before_lo"), d("before_hi")], [d("after_lo"), d("after_hi")]
    res = [d("res_lo"), d("res_hi")]
    old = [d("old_lo"), d("old_hi")]
    is_mem = opc["load"] + opc["store"]
    isdiv = mext * bit14
    sys_ = opc["system"]
    use_b, use_i = opc["op"] + opc["branch"] + opc["store"] + sys_, opc["imm"] + opc["load"] + opc["jalr"]
    for h, nm in enumerate(("lo", "hi")):
        C("u:" + nm, (1 - isdiv) * (u[h] - (is_mem * before[h] + (1 - is_mem) * a[h])))  # U: the word of a load / store, x[rs1] otherwise (a division row keeps its quotient there)
        C("v:" + nm, v[h] - (use_b * rs2[h] + use_i * immi[h]))               # V: x[rs2] or the I-immediate
    c0, c1, lt, eq, zinv, ob0, ob1 = d("c0"), d("c1"), d("lt"), d("eq"), d("zinv"), d("ob0"), d("ob1")
    for nm, x in (("c0", c0), ("c1", c1), ("lt", lt), ("eq", eq), ("ob0", ob0), ("ob1", ob1)):
        bit(x, nm)
    # the zero test looks at Z; on a division row at the divisor; on an ecall row at the register it counts down
    zero_of = z[0] + z[1] + isdiv * (v[0] + v[1] - z[0] - z[1]) + sys_ * (old[0] + old[1] - z[0] - z[1])
    C("eq:zero", eq * zero_of)
    C("eq:inv", zero_of * zinv - (1 - eq))
    differ = su + sv - 2 * su * sv
    C("lt", lt - (differ * su + (1 - differ) * c1))                    # signed U < V given the borrow c1 of U - V
    # --- the adder: X + Y = Z + 2^32 carry (halves, two carry bits), or backwards: Y + Z = X + 2^32 borrow
    sub_rr = alu * f3[0] * (opc["op"] * b30)
    sel_add = opc["jalr"] + opc["load"] + alu * f3[0] - sub_rr
    sel_sub = opc["branch"] + alu * (f3[2] + f3[3]) + sub_rr
    C("add:lo", sel_add * (a[0] + v[0] - z[0] - 65536 * c0))
    C("add:hi", sel_add * (a[1] + v[1] + c0 - z[1] - 65536 * c1))
    C("store:addr_lo", opc["store"] * (a[0] + imms[0] - z[0] - 65536 * c0))
    C("store:addr_hi", opc["store"] * (a[1] + imms[1] + c0 - z[1] - 65536 * c1))
    C("auipc:pc", opc["auipc"] * (w[0] + 65536 * w[1] - pc))
    C("auipc:lo", opc["auipc"] * (w[0] + immu[0] - z[0] - 65536 * c0))
    C("auipc:hi", opc["auipc"] * (w[1] + immu[1] + c0 - z[1] - 65536 * c1))
    C("sub:lo", sel_sub * (v[0] + z[0] - a[0] - 65536 * c0))
    C("sub:hi", sel_sub * (v[1] + z[1] + c0 - a[1] - 65536 * c1))
    # --- control flow
    link = opc["jal"] + opc["jalr"]
    C("next:plain", (live - link - opc["branch"] - opc["system"]) * (next_pc - pc - 4))
    C("next:jal", opc["jal"] * (next_pc - pc - imm_j))
    C("next:jalr", opc["jalr"] * (next_pc - (z[0] + 65536 * z[1] - ob0)))
    C("jalr:aligned", opc["jalr"] * ob1)
    taken = f3[0] * eq + f3[1] * (1 - eq) + f3[4] * lt + f3[5] * (1 - lt) + f3[6] * c1 + f3[7] * (1 - c1)
    C("next:branch", opc["branch"] * (next_pc - pc - 4 - taken * (imm_b - 4)))
    C("link", link * (w[0] + 65536 * w[1] - pc - 4))
    # --- loads and stores
    mem_act, mem_wr, top = d("mem_act"), d("mem_wr"), d("top")
    bit(mem_act, "mem_act")
    bit(mem_wr, "mem_wr")
    bit(top, "top")
    C("mem:wr_act", mem_wr * (1 - mem_act))
    C("top:bnd", (1 - bnd) * top)
    C("mem:act", (1 - sys_) * (mem_act - is_mem))
    C("mem:wr", (1 - sys_) * (mem_wr - opc["store"]))
    C("mem:addr", is_mem * (4 * d("addr3") + ob0 + 2 * ob1 - z[0] - 65536 * z[1]))
    narrow_h = opc["load"] * (f3[1] + f3[5]) + opc["store"] * f3[1]
    word = (opc["load"] + opc["store"]) * f3[2]
    C("mem:aligned_h", narrow_h * ob0)
    C("mem:aligned_w", word * (ob0 + ob1))
    sel_byte = [(1 - ob0) * (1 - ob1), ob0 * (1 - ob1), (1 - ob0) * ob1, ob0 * ob1]
    sb, sgn = d("sb"), d("sgn")
    bit(sgn, "sgn")
    C("sb", sb - lin(b, [(1, sel_byte[k] * ub[k]) for k in range(4)]))
    shalf = (1 - ob1) * u[0] + ob1 * u[1]
    ld = opc["load"]
    C("lb:lo", ld * f3[0] * (res[0] - sb - 0xFF00 * sgn))
    C("lb:hi", ld * f3[0] * (res[1] - 0xFFFF * sgn))
    C("lh:lo", ld * f3[1] * (res[0] - shalf))
    C("lh:hi", ld * f3[1] * (res[1] - 0xFFFF * sgn))
    C("lw:lo", ld * f3[2] * (res[0] - u[0]))
    C("lw:hi", ld * f3[2] * (res[1] - u[1]))
    C("lbu:lo", ld * f3[4] * (res[0] - sb))
    C("lbu:hi", ld * f3[4] * res[1])
    C("lhu:lo", ld * f3[5] * (res[0] - shalf))
    C("lhu:hi", ld * f3[5] * res[1])
    st = opc["store"]
    nb = [sel_byte[k] * vb[0] + (1 - sel_byte[k]) * ub[k] for k in range(4)]
    C("sb:lo", st * f3[0] * (after[0] - nb[0] - 256 * nb[1]))
    C("sb:hi", st * f3[0] * (after[1] - nb[2] - 256 * nb[3]))
    C("sh:lo", st * f3[1] * (after[0] - ((1 - ob1) * v[0] + ob1 * u[0])))
    C("sh:hi", st * f3[1] * (after[1] - (ob1 * v[0] + (1 - ob1) * u[1])))
    C("sw:lo", st * f3[2] * (after[0] - v[0]))
    C("sw:hi", st * f3[2] * (after[1] - v[1]))
    # --- the two spare range checks: what they hold, row kind by row kind (zero elsewhere)
    io = d("io")
    a31 = d("a31")
    bit(a31, "a31")
    dl = [d("dl%d" % k) for k in range(5)]
    dh = [d("dh%d" % k) for k in range(5)]
    C("aux0", d("aux0") - ((opc["jalr"] + is_mem) * (4 * z[1])                       # Z below 2^30: a jump target, a byte address
                           + io * (64 * z[1])                                          # the count of a transfer: below 2^26
                           + isdiv * (2 * (a[1] - 32768 * a31))                        # the dividend's sign bit
                           + bnd * (16 * dl[1])))                                      # a boundary address's high limb: 12 bits
    C("aux1", d("aux1") - ((link + opc["auipc"]) * (4 * w[1])                         # W below 2^30: the pc, the link
                           + ld * f3[0] * (512 * (sb - 128 * sgn))                     # LB: the selected byte's sign
                           + ld * f3[1] * (2 * (shalf - 32768 * sgn))                  # LH: the selected half's sign
                           + isdiv * (2 * (z[1] - 32768 * c1))                         # the remainder's sign bit
                           + io * (4 * v[1])                                           # a transfer's buffer: below 2^30
                           + bnd * (8 * w[1])))                                        # the gap to the previous boundary address: 29 bits
    # --- results of the register-writing instructions
    C("lui:lo", opc["lui"] * (res[0] - immu[0]))
    C("lui:hi", opc["lui"] * (res[1] - immu[1]))
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
        C("div:res_" + nm, isdiv * (res[h] - (bit13 * z[h] + (1 - bit13) * u[h])))  # DIV[U]: the quotient (U); REM[U]: the remainder (Z)
        C("ecall:res_" + nm, sys_ * (res[h] - z[h]))                    # what an ecall writes to a0 / a1: a range-checked word
        C("ecall:word_" + nm, sys_ * (after[h] - w[h]))                 # ... and to memory
        C("bnd:word_" + nm, bnd * (after[h] - z[h]))                    # the first value of an address is a 32-bit word
    C("slt:lo", alu * f3[2] * (res[0] - lt))
    C("slt:hi", alu * f3[2] * res[1])
    C("sltu:lo", alu * f3[3] * (res[0] - c1))
    C("sltu:hi", alu * f3[3] * res[1])
    # --- the multiplier: U (bytes) x M (byte limbs mb0..mb3) = Z + 2^32 W through four 16-bit positions.  Carries: a byte (looked
    # up) plus one or two radix-4 digits each; carry 0's byte is access 3's high timestamp limb (no multiplying row touches memory)
    p8, sx, sm, c3 = d("p8"), d("sx"), d("sm"), d("c3")
    mb = [d("mb%d" % k) for k in range(4)]
    shl, shr, mulsel = alu * f3[1], alu * f3[5], mext * (1 - bit14)
    sgnd = 1 - bit12                                                    # DIV / REM are signed, DIVU / REMU are not
    inv2 = lambda k: pow(pow(2, k, P), P - 2, P)
    pow_l = (1 + sh[0]) * (1 + 3 * sh[1]) * (1 + 15 * sh[2])
    pow_r = (1 + (inv2(1) - 1) * sh[0]) * (1 + (inv2(2) - 1) * sh[1]) * (1 + (inv2(4) - 1) * sh[2])
    C("p8", p8 - (shl * pow_l + 256 * (shr * pow_r)))
    q = [(1 - sh[3]) * (1 - sh[4]), sh[3] * (1 - sh[4]), (1 - sh[3]) * sh[4], sh[3] * sh[4]]
    for j in range(4):
        C("mb%d" % j, mb[j] - (mext * vb[j] + p8 * (shl * q[j] + shr * q[3 - j])))
    C("sx", sx - su * (mext * (f3[1] + f3[2]) + shr * b30 + isdiv * sgnd))
    C("sm", sm - sv * (mext * f3[1] + isdiv * sgnd))
    C("c3", c3 * (c3 + 1) * ((c3 + 2) * (c3 - 1)))
    for nm in ("ce0", "ce1a", "ce1b", "ce2"):
        digit(d(nm), nm)
    cm = [dh[3] + 256 * d("ce0"), d("cb1") + 256 * (d("ce1a") + 4 * d("ce1b")), d("cb2") + 256 * d("ce2")]
    s_ = [lin(b, [(1, ub[i] * mb[k - i]) for i in range(4) if 0 <= k - i < 4]) for k in range(7)]
    m_lo, m_hi = mb[0] + 256 * mb[1], mb[2] + 256 * mb[3]
    msel = mulsel + shl + shr
    C("mul:t0", msel * (s_[0] + 256 * s_[1] - z[0] - 65536 * cm[0]))
    C("mul:t1", msel * (s_[2] + 256 * s_[3] + cm[0] - z[1] - 65536 * cm[1]))
    C("mul:t2", msel * (s_[4] + 256 * s_[5] + cm[1] - sx * m_lo - sm * u[0] - w[0] - 65536 * (cm[2] - 4)))
    C("mul:t3", msel * (s_[6] + (cm[2] - 4) - sx * m_hi - sm * u[1] - w[1] - 65536 * c3))
    # --- division: U = quotient, V = divisor, Z = remainder, x[rs1] = dividend.  The same chain proves quotient x divisor + remainder =
    # dividend as 64-bit (sign-extended) integers; W = |divisor| - |remainder| - 1 is a range-checked word, so |remainder| < |divisor|;
    # a remainder other than 0 has the dividend's sign.  Division by zero: quotient all ones.  -2^31 / -1 (ovf): quotient = dividend
    dv, ovf, k0 = d("dv"), d("ovf"), d("k0")
    bit(ovf, "ovf")
    C("k0", (k0 + 1) * k0 * ((k0 - 1) * (k0 - 2)))
    C("dv", dv - isdiv * (1 - ovf))
    C("ovf:div", ovf * (1 - isdiv))
    C("ovf:signed", ovf * bit12)
    C("ovf:a_lo", ovf * a[0])
    C("ovf:a_hi", ovf * (a[1] - 0x8000))
    C("ovf:b_lo", ovf * (v[0] - 0xFFFF))
    C("ovf:b_hi", ovf * (v[1] - 0xFFFF))
    for h, nm in enumerate(("lo", "hi")):
        C("ovf:q_" + nm, ovf * (u[h] - a[h]))
        C("ovf:rem_" + nm, ovf * z[h])
        C("div0:q_" + nm, dv * eq * (u[h] - 0xFFFF))
    sr, sa, sb_ = c1 * sgnd, a31 * sgnd, sv * sgnd
    C("div:t0", dv * (s_[0] + 256 * s_[1] + z[0] - a[0] - 65536 * cm[0]))
    C("div:t1", dv * (s_[2] + 256 * s_[3] + cm[0] + z[1] - a[1] - 65536 * cm[1]))
    C("div:t2", dv * (s_[4] + 256 * s_[5] + cm[1] - sx * m_lo - sm * u[0] + 65535 * (sr - sa) - 65536 * (cm[2] - 4)))
    C("div:t3", dv * (s_[6] + (cm[2] - 4) - sx * m_hi - sm * u[1] + 65535 * (sr - sa) - 65536 * c3))
    C("div:rem_sign", dv * (sr - sa) * (z[0] + z[1]))
    cmp = dv * (1 - eq)
    C("div:less_lo", cmp * (w[0] + 1 + (1 - 2 * sr) * z[0] - (1 - 2 * sb_) * v[0] - 65536 * k0))
    C("div:less_hi", cmp * (w[1] + (1 - 2 * sr) * z[1] - (1 - 2 * sb_) * v[1] - 65536 * (sb_ - sr) + k0))
    # --- the five accesses.  Timestamp of an access of cycle c: 5 c + (1 fetch, 2 x[rs1], 3 x[rs2], 4 x[rd], 5 memory): larger than
    # the one it consumes by 1 + a 16-bit and an 8-bit limb (both looked up).  The tuples themselves are fractions of the running sum
    # (accesses 0, 1, 2 and 4 have no column for the consumed timestamp: it IS own - 1 - the limbs; access 3 keeps one, a boundary
    # row's consumed timestamp being free)
    act2, zrd = d("act2"), d("zrd")
    bit(act2, "act2")
    C("rd:zero", zrd * df["rd"])
    C("rd:inv", df["rd"] * d("inv_rd") - (1 - zrd))
    writes = opc["lui"] + opc["auipc"] + link + opc["load"] + opc["imm"] + opc["op"]
    C("rd:act", (1 - sys_) * (act2 - writes * (1 - zrd)))              # an instruction with a destination other than x0 writes it
    C("ordered:mem", mem_act * (5 * cycle + STAMP["mem"] - d("p3") - 1 - dl[3] - 65536 * dh[3]))
    keeps = 1 - mem_wr - bnd                                            # the word stays as it was unless written (or a boundary row)
    C("mem:keeps_lo", keeps * (after[0] - before[0]))
    C("mem:keeps_hi", keeps * (after[1] - before[1]))
    # --- ecalls: a7 (access 0) names the function (0 HALT, 1 READ_WORDS, 2 COMMIT, 3 CYCLES, 4 PAUSE), V = a0 (access 1).  The two
    # transfers count a1 down: while a1 = j > 0 the cycle moves word j - 1 of the buffer at a0, writes a1 = j - 1 and repeats; with
    # a1 = 0 it falls through.  Input words are the host's to say (range-checked); a COMMIT's word is named in the session sum
    f0, f1, f2, fn_cyc, cact = d("f0"), d("f1"), d("f2"), d("fn_cyc"), d("cact")
    for nm, x in (("f0", f0), ("f1", f1), ("f2", f2)):
        bit(x, nm)
    C("ecall:fn_lo", sys_ * (a[0] - f0 - 2 * f1 - 4 * f2))
    C("ecall:fn_hi", sys_ * a[1])
    C("ecall:fn_max", sys_ * f2 * (f0 + f1))
    C("ecall:fn_idle", (1 - sys_) * (f0 + f1 + f2))
    C("ecall:cycles", fn_cyc - sys_ * f0 * f1)
    active = io * (1 - eq)                                              # eq: a1 = 0
    C("ecall:commit", cact - active * f1)
    C("ecall:act2", sys_ * (act2 - io - fn_cyc))                        # the transfers write a1, CYCLES writes a0, HALT / PAUSE nothing
    C("ecall:count_lo", io * (old[0] - (1 - eq) - res[0] + 65536 * c0))  # a1 - 1 (a1 itself at 0) over the halves, c0 the borrow
    C("ecall:count_hi", io * (old[1] - c0 - res[1]))
    C("next:ecall", sys_ * (next_pc - pc - 4 + 4 * active))             # repeats while it moves, then falls through
    C("ecall:mem", sys_ * (mem_act - active))
    C("ecall:mem_wr", sys_ * (mem_wr - active * f0))                    # READ_WORDS writes memory, COMMIT reads it
    C("ecall:addr", active * (4 * d("addr3") - rs2[0] - 65536 * rs2[1] - 4 * (z[0] + 65536 * z[1])))
    C("ecall:buffer", io * (sh[0] + sh[1]))                             # a0: word-aligned (below 1 GiB: aux1)
    # --- boundary rows: one history per address.  The address is alo + 2^16 ahi (+ 2^28 for a register, then alo < 32), and exceeds the
    # previous boundary row's by 1 + a 29-bit gap: strictly increasing as integers.  The row names the segment that last held its
    # address (an earlier one) and, in a closing segment, the address's initial value: zero unless it is an image word
    C("bnd:addr", bnd * (d("addr3") - dl[0] - 65536 * dl[1] - (1 << 28) * top))
    C("bnd:reg_hi", bnd * top * dl[1])
    C("bnd:reg_lo", bnd * top * (dh[0] - 8 * dl[0]))
    gap = d("addr3") - d("addr3", 1) - 1 - w[0] - 65536 * w[1]
    C("bnd:order", not_first * bnd * prev_bnd * gap)
    fimg = d("fimg")
    bit(fimg, "fimg")
    C("fimg:bnd", (1 - bnd) * fimg)
    C("fimg:closing", fimg * (1 - gl(G_FIN)))
    for h, nm in enumerate(("lo", "hi")):
        C("bnd:init_" + nm, gl(G_FIN) * (bnd - fimg) * old[h])        # closing: an address outside the image starts at zero
    C("bnd:first_a", first * bnd * (d("addr3") - gl(G_ALO)))            # the public first and last boundary address (closing segments are chained by them)
    C("bnd:first_b", not_first * bnd * (1 - prev_bnd) * (d("addr3") - gl(G_ALO)))
    C("bnd:last_a", not_first * (1 - bnd) * prev_bnd * (d("addr3", 1) - gl(G_AHI)))
    C("bnd:last_b", last * bnd * (d("addr3") - gl(G_AHI)))
    # --- public inputs: the run starts at pc0 in cycle 0; the row after the last cycle (or the last row itself) pins the end.  A segment
    # without cycles (closing rows only) has cycles = 0
    C("first:kind", first * (1 - live - bnd))
    C("first:pc", first * live * (pc - gl(G_PC0)))
    C("first:cycle", first * cycle)
    C("first:none", first * bnd * gl(G_CYCLES))
    ended = not_first * (prev_live - live)                              # 1 on the first row that is not a cycle
    C("end:pc", ended * (d("next_pc", 1) - gl(G_PC1)))
    C("end:cycles", ended * (d("cycle", 1) + 1 - gl(G_CYCLES)))
    full = last * live
    C("full:pc", full * (next_pc - gl(G_PC1)))
    C("full:cycles", full * (cycle + 1 - gl(G_CYCLES)))
    fp_ = [d("f%d" % k, 1) for k in range(3)]
    term_prev = d("opc_system", 1) * (1 - fp_[0]) * (1 - fp_[1])        # the row before was a HALT (a7 = 0) or a PAUSE (a7 = 4)
    term_here = sys_ * (1 - f0) * (1 - f1)
    C("exit:last_cycle", not_first * term_prev * live)
    for tag, gate_, term, u2, lo, hi in (("end", ended, term_prev, fp_[2], d("rs2_lo", 1), d("rs2_hi", 1)), ("full", full, term_here, f2, rs2[0], rs2[1])):
        C("exit:%s_is" % tag, gate_ * (term - gl(G_TERM)))
        C("exit:%s_kind" % tag, gate_ * gl(G_TERM) * (1 + u2 - gl(G_KIND)))
        C("exit:%s_none" % tag, gate_ * (1 - gl(G_TERM)) * gl(G_KIND))
        C("exit:%s_lo" % tag, gate_ * (gl(G_TERM) * lo - gl(G_EXIT_LO)))
        C("exit:%s_hi" % tag, gate_ * (gl(G_TERM) * hi - gl(G_EXIT_HI)))
    # --- the running sums
    accum_constraints(b, E, fp4_mul_sym, accumulators(), first, cons)
    return b, cons


# ---------------------------------------------------------------------------------------------------------------- numpy checks
def code_columns(n):
    import numpy as np
    rows = np.arange(n, dtype=np.int64)
    small = rows < 65536
    t16 = np.where(small, rows, 0)
    tand = np.where(small, TAG_AND + rows + 65536 * ((rows & 255) & (rows >> 8)), TAG_AND)
    return [(rows == 0).astype(np.int64), (rows == n - 1).astype(np.int64), rows, np.zeros(n, dtype=np.int64), t16, tand]


_HASH = [[(0x9E3779B97F4A7C15 * (i + 1) ** 3 + 12345 * j) % P for i in range(8)] for j in range(2)]


def check_fractions(data, globals_, session_extra=()):
    """The log-derivative argument, exactly: the chained fractions must cancel as rational functions -- per distinct denominator the
    numerators sum to zero -- and so must the session fractions together with `session_extra` [(addr, lo, hi, t, numerator)] (what the
    other segments and the verifier contribute).  -> [(fraction name, rows, key)] of what does not cancel.  (Denominators are told
    apart by two 31-bit hashes of their forms' values: a collision would hide a mismatch with probability 2^-62.)"""
    import numpy as np
    n = data.shape[1]
    code = code_columns(n)
    chained, session = fractions()
    bad = []
    for group, extra in ((chained, ()), (session, session_extra)):
        keys, nums, rows, which = [], [], [], []
        for fi, f in enumerate(group):
            num = f.num.evaluate(data, code, globals_)
            r = np.nonzero(num)[0]
            if not len(r):
                continue
            parts = [lf.evaluate(data, code, globals_)[r] for _, lf in f.parts[1:]]
            h = [np.full(len(r), len(parts) * _HASH[j][7] % P, dtype=np.int64) for j in range(2)]
            for i, pv in enumerate(parts):
                for j in range(2):
                    h[j] = (h[j] + pv * _HASH[j][i]) % P
            keys.append(h[0] * (1 << 31) + h[1])
            nums.append(num[r])
            rows.append(r)
            which.append(np.full(len(r), fi, dtype=np.int64))
        for addr, lo, hi, t, num in extra:
            parts = [(-addr) % P, (-lo) % P, (-hi) % P, (-t) % P]
            h = [(4 * _HASH[j][7] + sum(pv * _HASH[j][i] for i, pv in enumerate(parts))) % P for j in range(2)]
            keys.append(np.array([h[0] * (1 << 31) + h[1]], dtype=np.int64))
            nums.append(np.array([num % P], dtype=np.int64))
            rows.append(np.array([-1], dtype=np.int64))
            which.append(np.array([-1], dtype=np.int64))
        if not keys:
            continue
        keys, nums, rows, which = map(np.concatenate, (keys, nums, rows, which))
        uniq, inv = np.unique(keys, return_inverse=True)
        # numerators are +-1 or multiplicities: sum them as signed integers (values above p / 2 stand for negatives)
        signed = np.where(nums > P // 2, nums - P, nums)
        net = np.zeros(len(uniq), dtype=np.int64)
        np.add.at(net, inv, signed)
        for u in np.nonzero(net % P)[0][:64]:
            members = np.nonzero(inv == u)[0]
            names = sorted({group[int(fi)].name if fi >= 0 else "extra" for fi in which[members]})
            bad.append(("|".join(names[:4]), sorted({int(x) for x in rows[members]})[:8], int(uniq[u])))
    return bad


def multiplicities(data, globals_):
    """fills m16 / mand of a witness [column, row] (canonical integers) from its lookups"""
    import numpy as np
    n = data.shape[1]
    code = code_columns(n)
    chained, _ = fractions()
    m = {TABLE_R16: np.zeros(n, dtype=np.int64), TABLE_AND: np.zeros(n, dtype=np.int64)}
    for f in chained:
        if not f.table:
            continue
        value = (-f.parts[1][1].evaluate(data, code, globals_)) % P
        num = f.num.evaluate(data, code, globals_)
        assert (num == 1).all()
        idx = value.astype(np.int64)
        if f.table == TABLE_AND:
            idx = idx - TAG_AND
            ok = (idx >= 0) & (idx < (1 << 24)) & (((idx & 255) & ((idx >> 8) & 255)) == (idx >> 16))
            idx = idx & 0xFFFF
        else:
            ok = (idx >= 0) & (idx < 65536)
        if not ok.all():
            raise ValueError("lookup %s: value not in its table at rows %s" % (f.name, np.nonzero(~ok)[0][:8].tolist()))
        np.add.at(m[f.table], idx, 1)
    data[COL["m16"]] = m[TABLE_R16]
    data[COL["mand"]] = m[TABLE_AND]


if __name__ == "__main__":
    ch, se = fractions()
    print("columns", len(TRACE_COLUMNS), "chained fractions", len(ch), "session", len(se), "logup words", len(logup_section()))
