"""The Poseidon2 sponge as a circuit component (the hash a recursion node's public digest is made with:
oracle/orc_core.c p2_mix / sponge_strided, csrc/ctx.cpp p2_hash_elems_host -- risc0-zkp 3.0.4 core/hash/poseidon2, t = 24, rate 16,
overwrite mode, zero padding, output = cells 0..7).

One permutation takes PERIOD = 30 consecutive rows, one linear or round step per row, and the schedule sits in CODE columns that repeat
with that period (kind 6: include/r0hip_circuit.h):

  row 0       absorb + the leading external layer   st = M_E(in[0..16) || st'[16..24))         (st' = the row before)
  rows 1..4   full rounds 0..3                       aux_j = (st'_j + rc_j)^3, st = M_E(aux_j^2 (st'_j + rc_j))
  rows 5..25  partial rounds 4..24                   aux_0 = (st'_0 + rc_0)^3, st = M_I(aux_0^2 (st'_0 + rc_0), st'_1, .., st'_23)
  rows 26..29 full rounds 25..28

x^7 is split as (x^3)^2 x with the cube in a column of its own, which keeps every constraint at degree <= 5 with its two gates.

DATA columns (SPONGE_DATA = 65): st[24], aux[24], in[16], act.  `act` is 1 from row 0 to the end of the last permutation and 0 after:
it may only fall behind a permutation's last row, and where it falls the state's first eight cells are the public digest words.  The
capacity before the first permutation is the last row's (wrap-around), held at zero.  CODE columns (SPONGE_CODE = 28): rc[24],
sel_mix, sel_full, sel_part, sel_last.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

P = 15 * 2**27 + 1
T, RATE, OUT = 24, 16, 8
HALF_FULL, PARTIAL = 4, 21
ROUNDS = 2 * HALF_FULL + PARTIAL
PERIOD = ROUNDS + 1
SPONGE_DATA = 2 * T + RATE + 1
SPONGE_CODE = T + 4
ST, AUX, IN, ACT = 0, T, 2 * T, 2 * T + RATE
RC, SEL_MIX, SEL_FULL, SEL_PART, SEL_LAST = 0, T, T + 1, T + 2, T + 3

_CONSTS = []


def consts():
    """-> (round constants [29*24], internal diagonal minus one [24]), canonical"""
    if not _CONSTS:
        import gen_poseidon2_consts
        _CONSTS.append(gen_poseidon2_consts.generate())
    return _CONSTS[0]


def m_ext(c):
    """the external layer on a list of 24 things that add: circ(2 M4, M4, .., M4), M4 by the paper's eight additions"""
    out = []
    for k in range(0, T, 4):
        a, b, d, e = c[k:k + 4]
        t0, t1 = a + b, d + e
        t2, t3 = b + b + t1, e + e + t0
        t1x2, t0x2 = t1 + t1, t0 + t0
        t4, t5 = t1x2 + t1x2 + t3, t0x2 + t0x2 + t2
        out += [t3 + t5, t5, t2 + t4, t4]
    col = [out[j] + out[4 + j] + out[8 + j] + out[12 + j] + out[16 + j] + out[20 + j] for j in range(4)]
    return [out[i] + col[i & 3] for i in range(T)]


def m_int(c, diag):
    total = c[0]
    for x in c[1:]:
        total = total + x
    return [total + c[i] * diag[i] for i in range(T)]


def schedule():
    """-> SPONGE_CODE columns of PERIOD canonical values: what the CODE columns of kind 6 repeat"""
    rc, _ = consts()
    cols = [[0] * PERIOD for _ in range(SPONGE_CODE)]
    cols[SEL_MIX][0] = 1
    for r in range(ROUNDS):
        full = r < HALF_FULL or r >= HALF_FULL + PARTIAL
        cols[SEL_FULL if full else SEL_PART][1 + r] = 1
        for j in range(T):
            cols[RC + j][1 + r] = rc[r * T + j]
    cols[SEL_LAST][PERIOD - 1] = 1
    return cols


def constraints(b, E, get_data, get_code, glob, first, last):
    """-> [(E, degree)]: get_data(col, back) / get_code(col, back) give E over the component's own columns; glob(j) the public digest
    word j; first / last the row indicators"""
    _, diag = consts()
    one = E.of(b, 1)
    st = [get_data(ST + j, 0) for j in range(T)]
    prev = [get_data(ST + j, 1) for j in range(T)]
    aux = [get_data(AUX + j, 0) for j in range(T)]
    inp = [get_data(IN + j, 0) for j in range(RATE)]
    act, act_prev = get_data(ACT, 0), get_data(ACT, 1)
    rc = [get_code(RC + j, 0) for j in range(T)]
    mix, full, part = get_code(SEL_MIX, 0), get_code(SEL_FULL, 0), get_code(SEL_PART, 0)
    last_round_prev = get_code(SEL_LAST, 1)
    out = []
    t = [prev[j] + rc[j] for j in range(T)]
    cube_gate = [act * (full + part)] + [act * full] * (T - 1)
    for j in range(T):
        out.append((cube_gate[j] * (aux[j] - t[j] * t[j] * t[j]), 5))
    s = [aux[j] * aux[j] * t[j] for j in range(T)]
    after_full = m_ext(s)
    after_part = m_int([s[0]] + prev[1:], diag)
    after_mix = m_ext(inp + prev[RATE:])
    for i in range(T):
        out.append((act * (st[i] - full * after_full[i] - part * after_part[i] - mix * after_mix[i]), 5))
    not_first = one - first
    fall = not_first * (act_prev - act)
    out.append((act * (one - act), 2))
    out.append((first * (one - act), 2))
    out.append((last * act, 2))
    out.append((not_first * (one - act_prev) * act, 3))
    out.append((fall * (one - last_round_prev), 3))
    for j in range(OUT):
        out.append((fall * (prev[j] - glob(j)), 3))
    for j in range(RATE, T):
        out.append((last * st[j], 2))
    for e, deg in out:
        assert e.deg <= deg <= 5, (e.deg, deg)
    return out


def permute(c):
    """the permutation on 24 canonical integers (a restatement for the witness below; tests pin it to the oracle's)"""
    rc, diag = consts()
    c = [x % P for x in m_ext(list(c))]
    for r in range(ROUNDS):
        if r < HALF_FULL or r >= HALF_FULL + PARTIAL:
            c = [x % P for x in m_ext([pow(c[j] + rc[r * T + j], 7, P) for j in range(T)])]
        else:
            c = [x % P for x in m_int([pow(c[0] + rc[r * T], 7, P)] + c[1:], diag)]
    return c


def witness(words, n_rows):
    """-> (SPONGE_DATA columns of n_rows canonical integers, the digest): the component's rows for the sponge over `words`"""
    rc, diag = consts()
    n_perm = max(1, (len(words) + RATE - 1) // RATE)
    assert n_perm * PERIOD < n_rows, "the words do not fit the trace"
    cols = [[0] * n_rows for _ in range(SPONGE_DATA)]
    state = [0] * T
    row = 0
    for q in range(n_perm):
        chunk = list(words[q * RATE:(q + 1) * RATE])
        chunk += [0] * (RATE - len(chunk))
        for j in range(RATE):
            cols[IN + j][row] = chunk[j] % P
        state = [x % P for x in m_ext(chunk + state[RATE:])]
        for j in range(T):
            cols[ST + j][row] = state[j]
        cols[ACT][row] = 1
        row += 1
        for r in range(ROUNDS):
            full = r < HALF_FULL or r >= HALF_FULL + PARTIAL
            lanes = range(T) if full else range(1)
            t = {j: (state[j] + rc[r * T + j]) % P for j in lanes}
            for j in lanes:
                cols[AUX + j][row] = pow(t[j], 3, P)
            s = [pow(t[j], 7, P) if j in t else state[j] for j in range(T)]
            state = [x % P for x in (m_ext(s) if full else m_int(s, diag))]
            for j in range(T):
                cols[ST + j][row] = state[j]
            cols[ACT][row] = 1
            row += 1
    return cols, state[:OUT]
