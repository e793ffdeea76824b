"""GPU parity: every C-ABI operation of libr0hip.so against the CPU oracle on the same seeded inputs, bit-exact
(all arithmetic on this path is integer: BabyBear words, digests, indices -- no tolerance anywhere).
Runs only on a real MI355X (`-m gpu`); nothing here reads /root/reference."""
import numpy as np
import pytest

from conftest import circuit_path

pytestmark = pytest.mark.gpu
P = 2013265921


def rnd(rng, n):
    return rng.integers(0, P, size=n, dtype=np.uint32)


@pytest.mark.parametrize("po2,count", [(1, 3), (4, 2), (9, 5), (12, 3), (13, 2), (14, 4), (17, 3), (20, 2), (21, 2), (22, 1)])
def test_interpolate_ntt(hal, orc, po2, count):
    rng = np.random.default_rng(100 + po2)
    x = rnd(rng, count << po2)
    buf = hal.copy_from(x)
    hal.batch_interpolate_ntt(buf, count, po2)
    assert np.array_equal(buf.to_host(), orc.batch_interpolate_ntt(x, count, po2))


@pytest.mark.parametrize("in_po2,expand,count", [(3, 0, 2), (5, 2, 3), (10, 2, 4), (12, 0, 2), (11, 2, 3), (14, 2, 2), (16, 2, 3), (20, 2, 1), (18, 0, 2), (19, 2, 2), (21, 0, 1), (22, 0, 1)])
def test_expand_into_evaluate_ntt(hal, orc, in_po2, expand, count):
    rng = np.random.default_rng(200 + in_po2)
    x = rnd(rng, count << in_po2)
    out = hal.alloc(count << (in_po2 + expand))
    hal.batch_expand_into_evaluate_ntt(out, hal.copy_from(x), count, in_po2, expand)
    assert np.array_equal(out.to_host(), orc.batch_expand_into_evaluate_ntt(x, count, in_po2, expand))


@pytest.mark.parametrize("po2,count", [(0, 3), (1, 2), (7, 3), (13, 5), (20, 2)])
def test_bit_reverse_and_zk_shift(hal, orc, po2, count):
    rng = np.random.default_rng(300 + po2)
    x = rnd(rng, count << po2)
    buf = hal.copy_from(x)
    hal.batch_bit_reverse(buf, count, po2)
    assert np.array_equal(buf.to_host(), orc.batch_bit_reverse(x, count, po2))
    hal.batch_bit_reverse(buf, count, po2)
    assert np.array_equal(buf.to_host(), x)  # involution
    hal.zk_shift(buf, count, po2)
    assert np.array_equal(buf.to_host(), orc.zk_shift(x, count, po2))


@pytest.mark.parametrize("rows,cols", [(64, 1), (256, 15), (256, 16), (300, 17), (1024, 48), (4096, 192), (512, 64), (1, 5)])
def test_hash_rows(hal, orc, rows, cols):
    rng = np.random.default_rng(rows + cols)
    m = rnd(rng, rows * cols)
    dig = hal.alloc(rows * 8)
    hal.hash_rows(dig, hal.copy_from(m), rows, cols)
    assert np.array_equal(dig.to_host(), orc.hash_rows(m, rows, cols))


def test_poseidon2_known_answer_vector_on_the_device(hal, orc):
    """KAT input (0..23) cannot enter the sponge directly (lanes 16..23 are capacity), so check the device against the
    vector through the pair hash: H(a || b) = first 8 words of the permutation of (a, b, 0^8), and the oracle -- which
    reproduces the published 24-word KAT -- must agree on the KAT's own first 16 words."""
    import json, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    kat = json.load(open(os.path.join(root, "tests/golden/poseidon2_kat_t24.json")))
    pair = np.array([orc.enc(v) for v in kat["input"][:16]], np.uint32)
    nodes = hal.copy_from(np.concatenate([np.zeros(16, np.uint32), pair]))  # nodes[2], nodes[3] hold the pair; output at nodes[1]
    hal.hash_fold(nodes, 1)
    got = nodes.to_host(8, 8)
    assert np.array_equal(got, orc.hash_pair(pair[:8], pair[8:]))
    # column form: a 16-column, 1-row matrix is one absorption of the same 16 words
    dig = hal.alloc(8)
    hal.hash_rows(dig, hal.copy_from(pair), 1, 16)
    assert np.array_equal(dig.to_host(), got)


def test_hash_rows_of_an_empty_matrix_is_one_permutation_of_zero(hal, orc):
    dig = hal.alloc(8 * 4)
    hal.hash_rows(dig, hal.alloc(4), 4, 0)
    want = orc.hash_elem_slice(np.zeros(0, np.uint32))
    assert np.array_equal(dig.to_host().reshape(4, 8), np.tile(want, (4, 1)))


@pytest.mark.parametrize("rows,cols", [(2, 3), (64, 16), (2048, 20), (1 << 14, 64), (1 << 16, 5), (1 << 18, 2)])  # levels on both fold kernels (cross-lane up to 8 K parents)
def test_hash_fold_and_merkle_build(hal, orc, rows, cols):
    rng = np.random.default_rng(rows)
    m = rnd(rng, rows * cols)
    nodes = hal.alloc(rows * 2 * 8)
    hal.zero(nodes)
    hal.merkle_build(nodes, hal.copy_from(m), rows, cols)
    want = orc.merkle_build(m, rows, cols)
    assert np.array_equal(nodes.to_host()[8:], want[8:])
    # one explicit fold level through the Hal entry point
    lvl = rnd(rng, rows * 2 * 8)
    nb = hal.copy_from(lvl)
    hal.hash_fold(nb, rows // 2)
    assert np.array_equal(nb.to_host(), orc.hash_fold(lvl, rows // 2))


def test_poseidon2_set_consts_changes_the_hash_and_default_table_matches_oracle(hal, orc):
    rng = np.random.default_rng(5)
    m = rnd(rng, 64 * 16)
    rc, diag = orc.poseidon2_consts()
    dig = hal.alloc(64 * 8)
    hal.hash_rows(dig, hal.copy_from(m), 64, 16)
    base = dig.to_host()
    rc2 = rc.copy()
    rc2[0] = (int(rc2[0]) + 1) % P
    hal.poseidon2_set_consts(rc2, diag)
    hal.hash_rows(dig, hal.copy_from(m), 64, 16)
    assert not np.array_equal(dig.to_host(), base)
    hal.poseidon2_set_consts(rc, diag)
    hal.hash_rows(dig, hal.copy_from(m), 64, 16)
    assert np.array_equal(dig.to_host(), base) and np.array_equal(base, orc.hash_rows(m, 64, 16))


@pytest.mark.parametrize("po2,cols", [(4, 3), (9, 6), (10, 4), (13, 7), (16, 5)])
def test_batch_evaluate_any(hal, orc, po2, cols):
    rng = np.random.default_rng(po2)
    coeffs = rnd(rng, cols << po2)
    xa, xb = rnd(rng, 4), rnd(rng, 4)
    which = rng.integers(0, cols, 9).astype(np.uint32)
    xs = np.concatenate([xa if k % 3 else xb for k in range(9)])
    out = hal.alloc(4 * 9)
    hal.batch_evaluate_any(hal.copy_from(coeffs), po2, which, xs, out)
    assert np.array_equal(out.to_host(), orc.batch_evaluate_any(coeffs, po2, which, xs))


@pytest.mark.parametrize("po2,cols,n_combo", [(5, 4, 2), (10, 12, 3), (14, 40, 5)])
def test_mix_poly_coeffs_and_sum(hal, orc, po2, cols, n_combo):
    rng = np.random.default_rng(po2 + cols)
    n = 1 << po2
    inp = rnd(rng, cols * n)
    combo_of = rng.integers(0, n_combo, cols).astype(np.uint32)
    start, mix = rnd(rng, 4), rnd(rng, 4)
    init = rnd(rng, 4 * n_combo * n)
    combos = hal.copy_from(init)
    hal.mix_poly_coeffs(combos, start, mix, hal.copy_from(inp), combo_of, po2)
    want = orc.mix_poly_coeffs(init, start, mix, inp, combo_of, po2)
    assert np.array_equal(combos.to_host(), want)
    out = hal.alloc(4 * n)
    hal.eltwise_sum_extelem(out, combos, n_combo, n)
    assert np.array_equal(out.to_host(), orc.eltwise_sum_extelem(want, n_combo, n))


@pytest.mark.parametrize("n_out", [1, 16, 256, 1 << 16])
def test_fri_fold(hal, orc, n_out):
    rng = np.random.default_rng(n_out)
    inp, mix = rnd(rng, 4 * 16 * n_out), rnd(rng, 4)
    out = hal.alloc(4 * n_out)
    hal.fri_fold(out, hal.copy_from(inp), mix, n_out)
    assert np.array_equal(out.to_host(), orc.fri_fold(inp, mix, n_out))


@pytest.mark.parametrize("n", [1, 2, 64, 256, 4096, 1 << 13, 1 << 17])
def test_prefix_products_and_poly_divide(hal, orc, n):
    rng = np.random.default_rng(n)
    v = rnd(rng, 4 * n)
    buf = hal.copy_from(v)
    hal.prefix_products(buf, n)
    assert np.array_equal(buf.to_host(), orc.prefix_products(v, n))
    z = rnd(rng, 4)
    buf = hal.copy_from(v)
    rem = hal.poly_divide(buf, n, z)
    q, want_rem = orc.poly_divide(v, n, z)
    assert np.array_equal(rem, want_rem) and np.array_equal(buf.to_host(), q)


def test_small_eltwise_ops(hal, orc):
    rng = np.random.default_rng(77)
    a, b = rnd(rng, 1000), rnd(rng, 1000)
    out = hal.alloc(1000)
    hal.eltwise_add_elem(out, hal.copy_from(a), hal.copy_from(b), 1000)
    assert np.array_equal(out.to_host().astype(np.int64), (a.astype(np.int64) + b) % P)
    hal.eltwise_copy_elem(out, hal.copy_from(b), 1000)
    assert np.array_equal(out.to_host(), b)
    z = a.copy()
    z[[0, 17, 999]] = 0xFFFFFFFF  # Elem::INVALID markers
    zb = hal.copy_from(z)
    hal.eltwise_zeroize_elem(zb, 1000)
    want_z = a.copy()
    want_z[[0, 17, 999]] = 0
    assert np.array_equal(zb.to_host(), want_z)
    g = hal.alloc(100)
    hal.gather_sample(g, hal.copy_from(a), 3, 100, 9)
    assert np.array_equal(g.to_host(), a[3::9][:100])
    into = hal.copy_from(np.zeros(50, np.uint32))
    offsets = np.array([4, 9, 2, 30, 31], np.uint32)
    vals = rnd(rng, 5)
    hal.scatter(into, hal.copy_from(np.array([0, 2, 5], np.uint32)), hal.copy_from(offsets), hal.copy_from(vals), 3)
    want = np.zeros(50, np.uint32)
    want[offsets] = vals
    assert np.array_equal(into.to_host(), want)


def test_argument_errors_are_reported_not_crashed(hal):
    import hyperfridge_r0_amd as r0
    small = hal.alloc(16)
    with pytest.raises(r0.R0HipError):
        hal.batch_interpolate_ntt(small, 4, 10)  # buffer too small
    with pytest.raises(r0.R0HipError):
        hal.batch_interpolate_ntt(small, 1, 30)  # size beyond the two-adicity the tables cover
    with pytest.raises(r0.R0HipError):
        hal.prefix_products(small, 3)
    with pytest.raises(r0.R0HipError):
        hal.load_circuit(np.zeros(10, np.uint32))


@pytest.mark.parametrize("name,po2", [("tiny", 9), ("small", 11)])
def test_witgen_accum_eval_check(hal, orc, name, po2):
    blob = np.fromfile(circuit_path(name), dtype=np.uint32)
    oc = orc.circuit(blob)
    gc = hal.load_circuit(blob)  # eval_check compiled in-process (hipRTC)
    assert gc.group_size == oc.group_size and gc.n_taps == oc.n_taps
    code, data, glob = hal.witgen(gc, po2, seed=5)
    ocode, odata, oglob = oc.witgen(po2, seed=5)
    assert np.array_equal(code.to_host(), ocode) and np.array_equal(data.to_host(), odata) and np.array_equal(glob, oglob)
    rng = np.random.default_rng(1)
    mix = rnd(rng, oc.n_mix)
    accum = hal.accum(gc, po2, code, data, mix)
    oaccum = oc.accum(po2, ocode, odata, mix)
    assert np.array_equal(accum.to_host(), oaccum)
    # eval_check on arbitrary (not even low-degree) group evaluations: pure arithmetic parity
    dom = 4 << po2
    ea, ec, ed = (rnd(rng, oc.group_size[g] * dom) for g in range(3))
    pm = rnd(rng, 4)
    check = hal.eval_check(gc, po2, hal.copy_from(ea), hal.copy_from(ec), hal.copy_from(ed), glob, mix, pm)
    assert np.array_equal(check.to_host(), oc.eval_check(po2, ea, ec, ed, glob, mix, pm))


@pytest.mark.parametrize("name,po2,seed", [("tiny", 9, 1), ("tiny", 12, 2), ("small", 10, 3), ("small", 13, 4)])
def test_prove_segment_seal_is_bit_identical_to_the_oracle_and_verifies(hal, orc, name, po2, seed):
    blob = np.fromfile(circuit_path(name), dtype=np.uint32)
    oc = orc.circuit(blob)
    gc = hal.load_circuit(blob)
    code, data, glob = hal.witgen(gc, po2, seed)
    seal = hal.prove_segment(gc, po2, code, data, glob)
    assert oc.verify(seal) == (0, "ok")
    want = oc.prove(po2, code.to_host(), data.to_host(), glob)
    assert seal.size == want.size and np.array_equal(seal, want)
    prof = hal.last_profile()
    assert [n for n, _ in prof][:3] == ["transcript_seed", "commit_code", "commit_data"]


def test_prove_segment_rejects_a_witness_that_breaks_the_taps(hal, orc):
    import hyperfridge_r0_amd as r0
    blob = np.fromfile(circuit_path("tiny"), dtype=np.uint32)
    oc = orc.circuit(blob)
    gc = hal.load_circuit(blob)
    code, data, glob = hal.witgen(gc, 9, 1)
    bad = data.to_host()
    bad[-1] = (int(bad[-1]) + 1) % P
    data.upload(bad)
    # the DEEP quotients still divide (they only depend on consistency of openings), so a seal comes out --
    # but the constraint identity at z fails and the verifier must say so
    seal = hal.prove_segment(gc, 9, code, data, glob)
    assert oc.verify(seal)[0] == 4


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_randomly_generated_circuits_prove_bit_identically(hal, orc, seed):
    """Fresh circuit structure per seed (taps, back-offsets, gates, accumulators differ): eval_check is compiled
    in-process, the seal must verify and equal the oracle's word for word."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import gen_circuit
    words, info = gen_circuit.generate(n_code=5 + seed % 3, n_data=18 + seed, n_acc=1 + seed % 3, n_free=5, n_pad=20 + seed,
                                       n_global=1 + seed % 3, seed=seed, cond_every=2 + seed % 3, comp=7)
    blob = np.array(words, dtype=np.uint32)
    oc = orc.circuit(blob)
    gc = hal.load_circuit(blob)
    po2 = 9 + seed % 3
    code, data, glob = hal.witgen(gc, po2, seed)
    seal = hal.prove_segment(gc, po2, code, data, glob)
    assert oc.verify(seal) == (0, "ok")
    assert np.array_equal(seal, oc.prove(po2, code.to_host(), data.to_host(), glob))


def test_prove_segment_mid_size_two_pass_ntt_split(hal, orc):
    """po2 = 17: the 2^17 / 2^19 transforms take the two-pass (contiguous + strided) radix-16 route inside the prover."""
    blob = np.fromfile(circuit_path("small"), dtype=np.uint32)
    oc = orc.circuit(blob)
    gc = hal.load_circuit(blob)
    code, data, glob = hal.witgen(gc, 17, 21)
    seal = hal.prove_segment(gc, 17, code, data, glob)
    assert oc.verify(seal) == (0, "ok")
    assert np.array_equal(seal, oc.prove(17, code.to_host(), data.to_host(), glob))


def test_wrapping_a_torch_tensor_is_zero_copy(orc, tmp_path):
    """PyTorch is only plumbing here: device memory it owns can be handed to the C ABI without a copy.  Run in a child
    process that initialises torch first (as bench.py does), so that one HIP runtime serves both."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rng = np.random.default_rng(9)
    po2, cols = 12, 3
    x = rnd(rng, cols << po2)
    np.save(tmp_path / "x.npy", x)
    script = """
import sys, numpy as np, torch
sys.path.insert(0, %r)
torch.cuda.init()
import hyperfridge_r0_amd as r0
x = np.load(%r)
t32 = torch.from_numpy(x.view(np.int32).copy()).cuda()
torch.cuda.synchronize()
hal = r0.Hal(0)
buf = hal.wrap(t32.data_ptr(), x.size)
hal.batch_interpolate_ntt(buf, %d, %d)
hal.sync()
np.save(%r, t32.cpu().numpy().view(np.uint32))
buf.free(); hal.close()
""" % (root, str(tmp_path / "x.npy"), cols, po2, str(tmp_path / "y.npy"))
    out = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    assert np.array_equal(np.load(tmp_path / "y.npy"), orc.batch_interpolate_ntt(x, cols, po2))


def test_two_contexts_prove_concurrently_and_agree(orc):
    """Two contexts on one device driven from two host threads (bench.py's in-flight mode) produce the same seals as one."""
    import threading
    import hyperfridge_r0_amd as r0
    blob = np.fromfile(circuit_path("small"), dtype=np.uint32)
    oc = orc.circuit(blob)
    hals = [r0.Hal(0), r0.Hal(0)]
    lanes = []
    for k, h in enumerate(hals):
        c = h.load_circuit(blob)
        code, data, glob = h.witgen(c, 12, 100 + k)
        lanes.append([h, c, code, data, glob, None])

    def work(ln):
        for _ in range(3):
            ln[5] = ln[0].prove_segment(ln[1], 12, ln[2], ln[3], ln[4])

    ts = [threading.Thread(target=work, args=(ln,)) for ln in lanes]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for ln in lanes:
        assert oc.verify(ln[5]) == (0, "ok")
        assert np.array_equal(ln[5], oc.prove(12, ln[2].to_host(), ln[3].to_host(), ln[4]))
        for obj in (ln[2], ln[3], ln[1]):
            obj.free()
        ln[0].close()


def _extreme(rng, n):
    """Words drawn from the corners of [0, p): the lazy-reduction bounds of the kernels are worst at p-1."""
    pool = np.array([0, 1, 2, P - 1, P - 2, (P - 1) // 2, (P + 1) // 2, 2**31 - 2**27, 0x0FFFFFFF, 0x70000000], dtype=np.uint32)
    return pool[rng.integers(0, pool.size, size=n)]


def test_extreme_field_values_through_every_arithmetic_kernel(hal, orc):
    rng = np.random.default_rng(4242)
    # Poseidon2: lane-per-row, lane-per-parent and the 24-lane cross-lane variant
    for rows, cols in ((512, 40), (4096, 16)):
        m = _extreme(rng, rows * cols)
        m[:rows] = P - 1  # a whole column at p-1
        dig = hal.alloc(rows * 8)
        hal.hash_rows(dig, hal.copy_from(m), rows, cols)
        assert np.array_equal(dig.to_host(), orc.hash_rows(m, rows, cols))
    for out_size in (1, 8, 1024, 2048, 8192):
        lvl = _extreme(rng, out_size * 4 * 8)
        nb = hal.copy_from(lvl)
        hal.hash_fold(nb, out_size)
        assert np.array_equal(nb.to_host(), orc.hash_fold(lvl, out_size))
    # NTTs (radix-16 and radix-2 routes), zk shift, bit reversal
    for po2, cols in ((10, 3), (13, 2), (17, 2)):
        x = _extreme(rng, cols << po2)
        buf = hal.copy_from(x)
        hal.batch_interpolate_ntt(buf, cols, po2)
        want = orc.batch_interpolate_ntt(x, cols, po2)
        assert np.array_equal(buf.to_host(), want)
        hal.zk_shift(buf, cols, po2)
        assert np.array_equal(buf.to_host(), orc.zk_shift(want, cols, po2))
        out = hal.alloc(cols << (po2 + 2))
        hal.batch_expand_into_evaluate_ntt(out, hal.copy_from(x), cols, po2, 2)
        assert np.array_equal(out.to_host(), orc.batch_expand_into_evaluate_ntt(x, cols, po2, 2))
    # extension-field streams with extreme operands (lazy 64-bit sums)
    n_out = 256
    inp, mix = _extreme(rng, 4 * 16 * n_out), np.array([P - 1] * 4, np.uint32)
    o = hal.alloc(4 * n_out)
    hal.fri_fold(o, hal.copy_from(inp), mix, n_out)
    assert np.array_equal(o.to_host(), orc.fri_fold(inp, mix, n_out))
    coeffs = _extreme(rng, 4 << 12)
    which, xs = np.array([0, 1, 2, 3], np.uint32), np.tile(np.array([P - 1, P - 1, P - 1, P - 1], np.uint32), 4)
    ev = hal.alloc(16)
    hal.batch_evaluate_any(hal.copy_from(coeffs), 12, which, xs, ev)
    assert np.array_equal(ev.to_host(), orc.batch_evaluate_any(coeffs, 12, which, xs))
    v = _extreme(rng, 4 * 8192)
    b = hal.copy_from(v)
    hal.prefix_products(b, 8192)
    assert np.array_equal(b.to_host(), orc.prefix_products(v, 8192))
    b = hal.copy_from(v)
    z = np.array([P - 1, P - 2, 1, 0], np.uint32)
    rem = hal.poly_divide(b, 8192, z)
    q, wr = orc.poly_divide(v, 8192, z)
    assert np.array_equal(rem, wr) and np.array_equal(b.to_host(), q)
    # constraint evaluation (generated code: lazy term sums) on extreme taps, globals and mixes
    blob = np.fromfile(circuit_path("small"), dtype=np.uint32)
    oc, gc = orc.circuit(blob), hal.load_circuit(blob)
    po2 = 9
    dom = 4 << po2
    ea, ec, ed = (_extreme(rng, oc.group_size[g] * dom) for g in range(3))
    glob, mixw, pm = _extreme(rng, oc.n_global), _extreme(rng, oc.n_mix), np.array([P - 1, P - 1, P - 1, P - 1], np.uint32)
    check = hal.eval_check(gc, po2, hal.copy_from(ea), hal.copy_from(ec), hal.copy_from(ed), glob, mixw, pm)
    assert np.array_equal(check.to_host(), oc.eval_check(po2, ea, ec, ed, glob, mixw, pm))


def test_hal_trait_operand_placement_forms_match_the_host_array_forms(hal, orc):
    """`which` / `xs` / `combos` as device buffers, `scatter` from host slices, `hash_fold(io, input_size, output_size)`: the
    operand placement of the risc0-zkp Hal trait (include/r0hip.h); results identical to the oracle's, as for the other forms."""
    import hyperfridge_r0_amd as r0
    rng = np.random.default_rng(5)
    po2, cols = 11, 7
    n = 1 << po2
    coeffs = rnd(rng, cols * n)
    which = np.array([0, 3, 3, 6, 1], np.uint32)
    xs = rnd(rng, 4 * which.size)
    out = hal.alloc(4 * which.size)
    hal.batch_evaluate_any_buf(hal.copy_from(coeffs), po2, hal.copy_from(which), hal.copy_from(xs), which.size, out)
    assert np.array_equal(out.to_host(), orc.batch_evaluate_any(coeffs, po2, which, xs))
    combo_of = np.array([0, 1, 1, 0, 2, 2, 1], np.uint32)
    ms, m = rnd(rng, 4), rnd(rng, 4)
    want = orc.mix_poly_coeffs(np.zeros(3 * n * 4, np.uint32), ms, m, coeffs, combo_of, po2)
    combos = hal.copy_from(np.zeros(3 * n * 4, np.uint32))
    hal.mix_poly_coeffs_buf(combos, ms, m, hal.copy_from(coeffs), hal.copy_from(combo_of), cols, po2)
    assert np.array_equal(combos.to_host(), want)
    into = hal.copy_from(np.zeros(50, np.uint32))
    offsets, vals = np.array([4, 9, 2, 30, 31], np.uint32), rnd(rng, 5)
    hal.scatter_slices(into, [1, 3, 5], offsets, vals)  # entries 1..4
    want = np.zeros(50, np.uint32)
    want[offsets[1:]] = vals[1:]
    assert np.array_equal(into.to_host(), want)
    with pytest.raises(r0.R0HipError, match="outside the destination"):
        hal.scatter_slices(into, [0, 5], np.array([4, 9, 2, 30, 50], np.uint32), vals)
    lvl = rnd(rng, 2 * 64 * 8)
    nb = hal.copy_from(lvl)
    hal.hash_fold_io(nb, 64, 32)
    assert np.array_equal(nb.to_host(), orc.hash_fold(lvl, 32))
    with pytest.raises(r0.R0HipError, match="2 \\* output_size"):
        hal.hash_fold_io(nb, 60, 32)
