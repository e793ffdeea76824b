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
  recursion  W=(24, 36, 161), 16 public inputs  the second circuit of SURVEY.md 8(a) a19 (lift/join at po2 = 18): the synthetic columns of
                                            W=(24, 8, 96) plus the Poseidon2 sponge component (tools/sponge_component.py) that computes the digest
                                            of what a node consumed -- public inputs 8..15 -- in-circuit
  image  W=(8, 30, 69)                      the program image's digest tied to its side of a session's memory argument (tools/image_circuit.py)
  trace  W=(40, 6, 128)                     columns = the executor's preflight rows (tools/trace_circuit.py): one contiguous run, every instruction's
                                            semantics, memory consistency, lookups, the session-wide memory argument
"""
import argparse
import os
import struct
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

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


def generate(n_code, n_data, n_acc, n_free, n_pad, n_global, seed, cond_every=5, comp=16, sponge=False):
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

    # the in-circuit sponge (tools/sponge_component.py) sits behind the synthetic columns of both groups
    n_code_all = n_code + (sponge_component.SPONGE_CODE if sponge else 0)
    n_data_all = n_data + (sponge_component.SPONGE_DATA if sponge else 0)
    b = Builder()
    for g, size in ((G_ACCUM, 4 * n_acc), (G_CODE, n_code_all), (G_DATA, n_data_all)):
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

    sponge_vals = []
    if sponge:
        assert n_global == 16, "the sponge's digest is public inputs 8..15"
        sponge_vals = [(e.v, deg) for e, deg in sponge_component.constraints(
            b, E, lambda col, back: E(b, b.get(G_DATA, n_data + col, back), 1), lambda col, back: E(b, b.get(G_CODE, n_code + col, back), 1),
            lambda j: E(b, b.glob(0, 8 + j), 0), E(b, first, 1), E(b, b.get(G_CODE, 1, 0), 1))]
        code_cols = code_cols + [(6, j) for j in range(sponge_component.SPONGE_CODE)]
        data_cols = data_cols + [(0, 0, 0, 0, 0)] * sponge_component.SPONGE_DATA

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
    for v, _ in sponge_vals:  # never behind a gate
        x = b.and_eqz(x, v)
    all_vals = all_vals + sponge_vals
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

    words = [MAGIC, 1, 9 if sponge else 7]
    words += section(SEC_INFO, list(struct.unpack("<4I", b"R0HIP_RECUR:v2__" if sponge else b"R0HIP_SYNTH:v1__")))
    words += section(SEC_GROUPS, [4 * n_acc, n_code_all, n_data_all])
    words += section(SEC_TAPS, [len(taps)] + [w for t in taps for w in t])
    words += section(SEC_GLOBALS, [n_global, 8 * n_acc] + global_cols)
    words += section(SEC_POLY, [len(steps), ret] + [w for s in steps for w in s])
    words += section(SEC_WITGEN, [n_code_all] + [w for cc in code_cols for w in cc] + [n_data_all] + [w for d in data_cols for w in d])
    words += section(SEC_ACCUM, [n_acc] + [w for a in acc_cols for w in a])
    if sponge:
        table = sponge_component.schedule()
        words += section(SEC_PERIODIC, [sponge_component.PERIOD, len(table)] + [v for col in table for v in col])
        words += section(SEC_SPONGE, [n_code, n_data, 8])
    info = {"taps": len(taps), "steps": len(steps), "constraints": len(all_vals), "mul_per_point": b.n_mul,
            "addsub_per_point": b.n_add, "groups": [4 * n_acc, n_code_all, n_data_all], "components": n_comp}
    return words, info


SEC_PERIODIC, SEC_SPONGE = 11, 12
import sponge_component  # noqa: E402


# ---- the trace circuit (tools/trace_circuit.py holds its columns, constraints and log-derivative argument) ------------------------
from trace_circuit import (OPCODES, TRACE_COLUMNS, TRACE_GLOBALS, TRACE_LATE, TRACE_MIX, REG_BASE, N_CODE, N_ACC, SEC_LATE, SEC_LOGUP,  # noqa: E402,F401
                           check_fractions, code_columns, logup_section, multiplicities)
import trace_circuit  # noqa: E402


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
    """-> (Builder, [(name, fp var, degree, touches ACCUM)])"""
    return trace_circuit.trace_constraints(Builder, E, lin, fp4_mul_sym)


def generate_trace():
    n_data, n_code, n_acc, n_global = len(TRACE_COLUMNS), N_CODE, N_ACC, TRACE_GLOBALS
    # first-row indicator, last-row indicator, row index, one seeded column, the two lookup tables (kinds 4 / 5: include/r0hip_circuit.h)
    code_cols = [(0, 0), (1, 0), (2, 0), (3, 3), (4, 0), (5, 0)]
    data_cols = [(0, 0, 0, 0, 0)] * n_data        # all free: the witness is the execution's
    b, cons = trace_constraints()
    x = b.true()
    for _, v, _, _ in cons:
        x = b.and_eqz(x, v)
    taps = sorted(b.taps)
    tap_index = {t: i for i, t in enumerate(taps)}
    steps = [(op_, tap_index[(a_[1], a_[2], a_[3])] if op_ == OP_GET else a_, b_, c_) for op_, a_, b_, c_ in b.steps]

    def section(tag, words):
        return [tag, len(words)] + list(words)

    words = [MAGIC, 1, 8]
    words += section(SEC_INFO, list(struct.unpack("<4I", b"R0HIP_TRACE:v5__")))
    words += section(SEC_GROUPS, [4 * n_acc, n_code, n_data])
    words += section(SEC_TAPS, [len(taps)] + [w for t in taps for w in t])
    words += section(SEC_GLOBALS, [n_global, TRACE_MIX])
    words += section(SEC_LATE, [TRACE_LATE])
    words += section(SEC_POLY, [len(steps), x] + [w for st in steps for w in st])
    words += section(SEC_WITGEN, [n_code] + [w for cc in code_cols for w in cc] + [n_data] + [w for dc in data_cols for w in dc])
    words += section(SEC_LOGUP, logup_section())
    info = {"taps": len(taps), "steps": len(steps), "constraints": len(cons), "mul_per_point": b.n_mul, "addsub_per_point": b.n_add,
            "groups": [4 * n_acc, n_code, n_data], "columns": len(TRACE_COLUMNS)}
    return words, info


_TRACE_CACHE = []


def check_trace_rows(data, globals_, first_only=True, session_extra=None):
    """Evaluate every DATA / CODE constraint of the trace circuit on a witness: data[column][row] canonical integers (numpy int64),
    globals_ the TRACE_GLOBALS public inputs as canonical integers.  -> [(constraint name, rows where it does not vanish)], followed by
    what the log-derivative argument would not accept: [(fraction name, rows)] whose tuples / looked-up values do not cancel (the
    session fractions are checked only when `session_extra` -- the other side's tuples -- is given).  A development and test aid:
    tells WHICH constraint a witness breaks, where the prover only says that one does."""
    import numpy as np
    if not _TRACE_CACHE:
        _TRACE_CACHE.append(trace_constraints())
    b, cons = _TRACE_CACHE[0]
    n = data.shape[1]
    code = code_columns(n)
    wanted = {v for _, v, _, accum in cons if not accum}
    vals, fp = {}, 0
    zero = np.zeros(n, dtype=np.int64)
    # only what the non-ACCUM constraints reach is evaluated (the accumulator polynomials are covered by check_fractions)
    need = set()
    stack = list(wanted)
    index = []
    for op, a_, b_, c_ in b.steps:
        if op in (OP_TRUE, OP_AND_EQZ, OP_AND_COND):
            continue
        index.append((op, a_, b_, c_))
    while stack:
        v = stack.pop()
        if v in need:
            continue
        need.add(v)
        op, a_, b_, c_ = index[v]
        if op in (OP_ADD, OP_SUB, OP_MUL):
            stack += [a_, b_]
    for fp, (op, a_, b_, c_) in enumerate(index):
        if fp not in need:
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
    bad = []
    for name, v, _, accum in cons:
        if accum:
            continue
        where = np.nonzero(vals[v])[0]
        if len(where):
            bad.append((name, where[:8].tolist() if first_only else where.tolist()))
    for name, rows, _ in check_fractions(data, globals_, session_extra if session_extra is not None else ()):
        if (name.startswith("session:") or name == "extra") and session_extra is None:
            continue
        bad.append(("sum:" + name, rows))
    return bad


SHAPES = {
    "tiny": dict(n_code=4, n_data=12, n_acc=2, n_free=4, n_pad=6, n_global=2, seed=1, comp=6),
    # 8 public inputs: enough to name a receipt claim (r0h_claim_globals), as `bench` can
    "small": dict(n_code=8, n_data=40, n_acc=4, n_free=8, n_pad=60, n_global=8, seed=2, comp=10),
    "bench": dict(n_code=16, n_data=192, n_acc=12, n_free=24, n_pad=2600, n_global=8, seed=3, comp=16),
    # a smaller trace (proved at po2 = 18) whose 16 public inputs carry the two 8-word digests a lift/join step stands for
    # (hyperfridge-r0_amd/recursion.py); the second digest is the output of the in-circuit sponge.  It does not verify seals in-circuit:
    # risc0's recursion circuit is not reproducible here.
    "recursion": dict(n_code=8, n_data=96, n_acc=6, n_free=20, n_pad=900, n_global=16, seed=4, comp=12, sponge=True),
}


def generate_image():
    import image_circuit
    return image_circuit.generate(Builder, E, fp4_mul_sym, OP_GET)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("shape", choices=sorted(SHAPES) + ["trace", "image"])
    ap.add_argument("out")
    args = ap.parse_args()
    words, info = generate_trace() if args.shape == "trace" else generate_image() if args.shape == "image" else generate(**SHAPES[args.shape])
    with open(args.out, "wb") as f:
        f.write(struct.pack("<%dI" % len(words), *words))
    print(args.shape, info, "words", len(words))


if __name__ == "__main__":
    sys.exit(main())
