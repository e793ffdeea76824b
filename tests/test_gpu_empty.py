"""Zero-sized calls through the C ABI: every streaming operation must treat an empty range as a no-op (or refuse it with an
error string) -- never launch an empty grid, never touch memory.  risc0's Hal is called with empty slices in a few places
(e.g. an accumulator group of width zero)."""
import numpy as np
import pytest

import hyperfridge_r0_amd as r0

pytestmark = pytest.mark.gpu
P = 2013265921


def _ok_or_clean_error(fn):
    try:
        fn()
    except r0.R0HipError as e:
        assert "invalid configuration" not in str(e).lower() and "hip" not in str(e).lower().split(":")[0], str(e)


def test_zero_length_operations_leave_memory_untouched(hal):
    rng = np.random.default_rng(3)
    ref = rng.integers(0, P, 4096, dtype=np.uint32)
    a, b, out = hal.copy_from(ref), hal.copy_from(ref), hal.copy_from(ref)
    calls = [
        lambda: hal.eltwise_add_elem(out, a, b, 0),
        lambda: hal.eltwise_copy_elem(out, a, 0),
        lambda: hal.eltwise_zeroize_elem(out, 0),
        lambda: hal.eltwise_sum_extelem(out, a, 4, 0),
        lambda: hal.gather_sample(out, a, 0, 0, 1),
        lambda: hal.batch_interpolate_ntt(out, 0, 10),
        lambda: hal.batch_expand_into_evaluate_ntt(out, a, 0, 8, 2),
        lambda: hal.batch_bit_reverse(out, 0, 10),
        lambda: hal.zk_shift(out, 0, 10),
        lambda: hal.prefix_products(out, 0),
        lambda: hal.fri_fold(out, a, np.array([1, 2, 3, 4], np.uint32), 0),
        lambda: hal.batch_evaluate_any(a, 10, np.zeros(0, np.uint32), np.zeros(0, np.uint32), out),
    ]
    for k, call in enumerate(calls):
        _ok_or_clean_error(call)
        hal.sync()
        assert np.array_equal(out.to_host(), ref) and np.array_equal(a.to_host(), ref), "call %d wrote through an empty range" % k
    # the empty SUM is the zero polynomial: n extension elements of output, nothing read
    hal.eltwise_sum_extelem(out, a, 0, 16)
    got = out.to_host()
    assert not got[:64].any() and np.array_equal(got[64:], ref[64:])
    out = hal.copy_from(ref)
    # and the context is still healthy afterwards
    hal.eltwise_add_elem(out, a, b, 4096)
    assert np.array_equal(out.to_host(), ((ref.astype(np.uint64) * 2) % P).astype(np.uint32))
