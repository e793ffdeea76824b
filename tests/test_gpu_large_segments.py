"""Segments above risc0's default size: 2^21 .. 2^24 rows (R0H_MAX_PO2 = 24, risc0's own maximum), i.e. evaluation domains of
2^23 .. 2^26 points -- what 288 GB of HBM has room for.  The transforms take a third level there (csrc/ntt.hip: split16_for);
everything is compared with the CPU oracle bit for bit, as at the default size.  `-m gpu` only."""
import numpy as np
import pytest

from conftest import circuit_path

pytestmark = pytest.mark.gpu
P = 2013265921


def rnd(rng, n):
    return rng.integers(0, P, size=n, dtype=np.uint32)


@pytest.mark.parametrize("po2,count", [(23, 2), (24, 2), (25, 1), (26, 1)])
def test_interpolate_ntt_above_two_passes(hal, orc, po2, count):
    rng = np.random.default_rng(700 + po2)
    x = rnd(rng, count << po2)
    buf = hal.copy_from(x)
    hal.batch_interpolate_ntt(buf, count, po2)
    assert np.array_equal(buf.to_host(), orc.batch_interpolate_ntt(x, count, po2))


@pytest.mark.parametrize("in_po2,expand,count", [(21, 2, 2), (22, 2, 2), (23, 2, 1), (24, 2, 1), (23, 0, 1), (24, 0, 2), (26, 0, 1)])
def test_expand_into_evaluate_ntt_above_two_passes(hal, orc, in_po2, expand, count):
    rng = np.random.default_rng(800 + in_po2 + expand)
    x = rnd(rng, count << in_po2)
    out = hal.alloc(count << (in_po2 + expand))
    hal.batch_expand_into_evaluate_ntt(out, hal.copy_from(x), count, in_po2, expand)
    assert np.array_equal(out.to_host(), orc.batch_expand_into_evaluate_ntt(x, count, in_po2, expand))


def test_forward_and_inverse_are_inverse_at_every_large_size(hal):
    """size-independent property at the sizes the oracle is slow for with many columns: evaluate(interpolate(x)) == x"""
    rng = np.random.default_rng(5)
    for po2, count in [(23, 5), (24, 3), (25, 3), (26, 2)]:
        x = rnd(rng, count << po2)
        buf = hal.copy_from(x)
        hal.batch_interpolate_ntt(buf, count, po2)
        out = hal.alloc(count << po2)
        hal.batch_expand_into_evaluate_ntt(out, buf, count, po2, 0)
        assert np.array_equal(out.to_host(), x), po2


@pytest.mark.parametrize("po2,count", [(23, 2), (24, 1), (26, 1)])
def test_bit_reverse_and_zk_shift_large(hal, orc, po2, count):
    rng = np.random.default_rng(900 + po2)
    x = rnd(rng, count << po2)
    buf = hal.copy_from(x)
    hal.batch_bit_reverse(buf, count, po2)
    assert np.array_equal(buf.to_host(), orc.batch_bit_reverse(x, count, po2))
    hal.batch_bit_reverse(buf, count, po2)
    assert np.array_equal(buf.to_host(), x)
    hal.zk_shift(buf, count, po2)
    assert np.array_equal(buf.to_host(), orc.zk_shift(x, count, po2))


@pytest.mark.parametrize("po2,count", [(5, 3), (10, 2), (14, 2), (20, 2), (22, 1), (23, 2), (24, 2), (25, 1), (26, 1)])
def test_interpolate_with_the_coset_shift_fused_equals_the_two_calls(hal, orc, po2, count):
    """what the sequencer issues for every group: one-pass, two-pass, ROU[26]-table and three-level transforms all carry the shift"""
    rng = np.random.default_rng(950 + po2)
    x = rnd(rng, count << po2)
    buf = hal.copy_from(x)
    hal.batch_interpolate_ntt_zk_shift(buf, count, po2)
    assert np.array_equal(buf.to_host(), orc.zk_shift(orc.batch_interpolate_ntt(x, count, po2), count, po2))


def test_more_columns_than_one_launch_takes(hal, orc):
    """The blocks of a large transform count as columns of its inner passes, so launches are cut at 32,768 columns: the same cut
    with plain columns -- 33,000 of 2^8 (one pass) and of 2^16 (two passes) -- checked on columns either side of it."""
    rng = np.random.default_rng(6)
    for po2, count in [(8, 33000), (16, 33000)]:
        x = rnd(rng, count << po2)
        buf = hal.copy_from(x)
        hal.batch_interpolate_ntt(buf, count, po2)
        got = buf.to_host().reshape(count, 1 << po2)
        out = hal.alloc(count << po2)
        hal.batch_expand_into_evaluate_ntt(out, buf, count, po2, 0)
        assert np.array_equal(out.to_host(), x)
        for c in (0, 1, 32767, 32768, 32769, count - 1):
            col = x[c << po2:(c + 1) << po2]
            assert np.array_equal(got[c], orc.batch_interpolate_ntt(col, 1, po2)), (po2, c)


@pytest.mark.parametrize("name,po2,seed", [("tiny", 21, 31), ("tiny", 22, 32)])
def test_large_segment_seal_is_bit_identical_to_the_oracle_and_bound_to_its_control_root(hal, orc, name, po2, seed):
    """2^23-point domains take two passes with 2^10-row tiles and the ROU[26] tables, 2^24 and up three levels (the oracle needs
    about a minute per seal here: the larger sizes below go through its verifier only)"""
    import hyperfridge_r0_amd as r0
    blob = np.fromfile(circuit_path(name), dtype=np.uint32)
    oc = orc.circuit(blob)
    gc = hal.load_circuit(blob)
    code, data, glob = hal.witgen(gc, po2, seed)
    seal = hal.prove_segment(gc, po2, code, data, glob)
    root = hal.code_root(gc, po2, code)
    assert oc.verify(seal, code_root=root) == (0, "ok")
    want = oc.prove(po2, code.to_host(), data.to_host(), glob)
    assert seal.size == want.size and np.array_equal(seal, want)
    verdict, why, got_po2 = r0.verify_seal(blob, seal, code_root=root)
    assert (verdict, got_po2) == (0, po2), why


@pytest.mark.parametrize("name,po2,seed", [("small", 22, 41), ("small", 23, 42), ("tiny", 24, 43)])
def test_largest_segments_are_accepted_by_both_verifiers(hal, orc, name, po2, seed):
    import hyperfridge_r0_amd as r0
    blob = np.fromfile(circuit_path(name), dtype=np.uint32)
    oc = orc.circuit(blob)
    gc = hal.load_circuit(blob)
    code, data, glob = hal.witgen(gc, po2, seed)
    seal = hal.prove_segment(gc, po2, code, data, glob)
    root = hal.code_root(gc, po2, code)
    assert oc.verify(seal, code_root=root) == (0, "ok")
    verdict, why, got_po2 = r0.verify_seal(blob, seal, code_root=root)
    assert (verdict, got_po2) == (0, po2), why
    wrong = root.copy()
    wrong[0] ^= 1
    assert oc.verify(seal, code_root=wrong)[0] == 10 and r0.verify_seal(blob, seal, code_root=wrong)[0] == 10


def test_bench_circuit_segment_of_four_million_rows(hal, orc):
    """256 columns x 2^22 rows: ~32 GiB resident on the device.  The oracle's prover would take minutes and tens of GiB here;
    its verifier (and the product's) take the seal, bound to the control root of that size."""
    import hyperfridge_r0_amd as r0
    blob = np.fromfile(circuit_path("bench"), dtype=np.uint32)
    oc = orc.circuit(blob)
    gc = hal.load_circuit(blob, circuit_path("bench").replace(".r0c", ".evalcheck.hsaco"))
    po2 = 22
    code, data, glob = hal.witgen(gc, po2, 77)
    seal = hal.prove_segment(gc, po2, code, data, glob)
    root = hal.code_root(gc, po2, code)
    assert oc.verify(seal, code_root=root) == (0, "ok")
    verdict, why, got_po2 = r0.verify_seal(blob, seal, code_root=root)
    assert (verdict, got_po2) == (0, po2), why
    flipped = seal.copy()
    flipped[seal.size // 2] ^= 1
    assert oc.verify(flipped, code_root=root)[0] != 0
