"""The image circuit (ProtocolInfo "R0HIP_IMAGE:v1__"): ties a program image's 32-byte digest to what the image contributes to a
session's memory argument, so that a verifier who holds the image id alone -- `receipt.verify(image_id)`,
verifier/src/main.rs:124-126 -- can check the balance r0h_receipt_verify_elf checks with the ELF in hand.

The image is the ELF's words in address order, (word index, word).  Its digest D is the Poseidon2 sponge (the library's one: rate 16,
overwrite mode) over blocks of 16 words, one block per permutation:

    addr_0, lo_0, hi_0, .. addr_3, lo_3, hi_3, mask, 0, 0, 0          mask = sum of 2^j over the tuples j < 4 that are there

(the tuples there are a prefix of the block, the others are zero; only the last block may be short; an image without words is one
block of zeros).  D is the root of the initial memory state the image id names (csrc/rv32im.cpp elf_image).

The circuit runs the sponge component (tools/sponge_component.py) over those blocks -- its result tied to public inputs 0..7 -- and,
on the row that absorbs a block, adds v_j / (alpha_g - addr_j - gamma lo_j - gamma^2 hi_j - gamma^3 TAG_IMG) for each tuple to a
running sum whose total is public (inputs 24..27): exactly the fractions the trace circuit's closing rows name for image words,
under the session's challenge (inputs 8..23; late, like the trace circuit's).  v_j are the mask's bits.  A prover who does not know
words hashing to D cannot make the proof; with them, the total is the image's side of the balance.

W = (8 ACCUM, 30 CODE, 69 DATA); public inputs: D (8), challenge (16, late), total (4, late).
"""
import os
import struct
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import sponge_component as sc
from trace_circuit import G_ACCUM, G_CODE, G_DATA, LF, ONE, SEC_LATE, SEC_LOGUP, TAG_IMG, Fraction, accum_constraints

P = 15 * 2**27 + 1
TUPLES = 4                                   # per block
MASK_AT = 3 * TUPLES                         # the block's word that holds the mask
N_CODE = 2 + sc.SPONGE_CODE                  # first row, last row, the sponge's schedule
SPONGE_AT = 0                                # DATA: the sponge's 65 columns, then the four flags
V_AT = sc.SPONGE_DATA
N_DATA = sc.SPONGE_DATA + TUPLES
G_DIGEST, G_GAMMA, G_SUM = 0, 8, 24
N_GLOBALS, N_LATE, N_MIX = 28, 20, 4
MIN_PO2 = 6


def col(i):
    return LF.col(i)


def in_col(k):
    return col(SPONGE_AT + sc.IN + k)


def fractions():
    """the chain's (a link that adds nothing: the argument's engine wants one) and the image accumulator's"""
    nothing = [Fraction("nothing:%d" % k, 0, [(("one",), ONE)]) for k in range(4)]
    image = []
    for j in range(TUPLES):
        image.append(Fraction("image:%d" % j, col(V_AT + j),
                              [(("glob", G_GAMMA), ONE), (("one",), -in_col(3 * j)), (("glob", G_GAMMA + 4), -in_col(3 * j + 1)),
                               (("glob", G_GAMMA + 8), -in_col(3 * j + 2)), (("glob", G_GAMMA + 12), -LF.of(TAG_IMG))]))
    return [(nothing, None), (image, G_SUM)]


def logup_section():
    accs = fractions()
    words = [len(accs), 0]
    for fr, final in accs:
        words += [len(fr), 0xFFFFFFFF if final is None else final]
        for f in fr:
            words += f.words()
    return words


def constraints(builder_cls, E, fp4_mul_sym):
    b = builder_cls()
    accs = fractions()
    for g, size in ((G_ACCUM, 4 * len(accs)), (G_CODE, N_CODE), (G_DATA, N_DATA)):
        for c in range(size):
            b.taps.add((g, c, 0))
    cons = []
    first, last = E(b, b.get(G_CODE, 0, 0), 1), E(b, b.get(G_CODE, 1, 0), 1)
    d = lambda c, back=0: E(b, b.get(G_DATA, c, back), 1)
    code = lambda c, back=0: E(b, b.get(G_CODE, 2 + c, back), 1)
    for k, (e, deg) in enumerate(sc.constraints(b, E, lambda c, back: d(SPONGE_AT + c, back), code, lambda j: E(b, b.glob(0, G_DIGEST + j), 0), first, last)):
        cons.append(("sponge:%d" % k, e.v, deg, False))
    act, mix = d(SPONGE_AT + sc.ACT), code(sc.SEL_MIX)
    v = [d(V_AT + j) for j in range(TUPLES)]
    for j in range(TUPLES):
        cons.append(("v%d:bit" % j, (v[j] * (v[j] - 1)).v, 2, False))
        cons.append(("v%d:absorbing" % j, (v[j] * (1 - mix)).v, 2, False))   # a tuple counts on the row that absorbs its block
        cons.append(("v%d:active" % j, (v[j] * (1 - act)).v, 2, False))      # ... of a permutation that is part of the sponge
    mask = v[0] + 2 * v[1] + 4 * v[2] + 8 * v[3]
    cons.append(("mask", (act * mix * (d(SPONGE_AT + sc.IN + MASK_AT) - mask)).v, 3, False))  # the flags are the hashed mask's bits
    accum_constraints(b, E, fp4_mul_sym, accs, first, cons)
    return b, cons


def generate(builder_cls, E, fp4_mul_sym, op_get):
    b, cons = constraints(builder_cls, E, fp4_mul_sym)
    x = b.true()
    for _, var, _, _ in cons:
        x = b.and_eqz(x, var)
    taps = sorted(b.taps)
    tap_index = {t: i for i, t in enumerate(taps)}
    steps = [(op_, tap_index[(a_[1], a_[2], a_[3])] if op_ == op_get else a_, b_, c_) for op_, a_, b_, c_ in b.steps]

    def section(tag, words):
        return [tag, len(words)] + list(words)

    code_cols = [(0, 0), (1, 0)] + [(6, j) for j in range(sc.SPONGE_CODE)]
    table = sc.schedule()
    words = [0x31433052, 1, 10]
    words += section(7, list(struct.unpack("<4I", b"R0HIP_IMAGE:v1__")))
    words += section(1, [4 * len(fractions()), N_CODE, N_DATA])
    words += section(2, [len(taps)] + [w for t in taps for w in t])
    words += section(3, [N_GLOBALS, N_MIX])
    words += section(SEC_LATE, [N_LATE])
    words += section(4, [len(steps), x] + [w for st in steps for w in st])
    words += section(5, [N_CODE] + [w for cc in code_cols for w in cc] + [N_DATA] + [0] * (5 * N_DATA))
    words += section(SEC_LOGUP, logup_section())
    words += section(11, [sc.PERIOD, len(table)] + [v for column in table for v in column])
    words += section(12, [2, SPONGE_AT, G_DIGEST])
    info = {"taps": len(taps), "steps": len(steps), "constraints": len(cons), "mul_per_point": b.n_mul, "addsub_per_point": b.n_add,
            "groups": [4 * len(fractions()), N_CODE, N_DATA]}
    return words, info


# ---- the witness, restated for the tests (canonical integers; the library's is csrc/claim.cpp image_witness)
def blocks(image):
    """image: [(word index, word)] in address order -> the sponge's blocks of 16 canonical words"""
    out = []
    for at in range(0, max(1, len(image)), TUPLES):
        part = image[at:at + TUPLES]
        blk = []
        for addr, word in part:
            blk += [addr, word & 0xFFFF, word >> 16]
        blk += [0] * (MASK_AT - len(blk)) + [(1 << len(part)) - 1, 0, 0, 0]
        out.append(blk)
    return out


def witness(image, n_rows):
    """-> (N_DATA columns of n_rows canonical integers, digest)"""
    stream = [w for blk in blocks(image) for w in blk]
    cols, digest = sc.witness(stream, n_rows)
    flags = [[0] * n_rows for _ in range(TUPLES)]
    for q, blk in enumerate(blocks(image)):
        for j in range(TUPLES):
            flags[j][q * sc.PERIOD] = (blk[MASK_AT] >> j) & 1
    return cols + flags, digest
