"""ctypes binding of oracle/liborc.so -- the CPU restatement used as the checker (tests, smoke, cpu_baseline only)."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "liborc.so")
P = 2013265921
_vp, _u32, _u64, _sz = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_size_t


def _ptr(a):
    return a.ctypes.data_as(_vp)


def u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


class Oracle:
    def __init__(self, lib):
        self.L = L = lib
        L.orc_fp_mul.restype = _u32; L.orc_fp_mul.argtypes = [_u32, _u32]
        L.orc_fp_enc.restype = _u32; L.orc_fp_enc.argtypes = [_u32]
        L.orc_fp_dec.restype = _u32; L.orc_fp_dec.argtypes = [_u32]
        L.orc_fp_inv.restype = _u32; L.orc_fp_inv.argtypes = [_u32]
        L.orc_fp_pow.restype = _u32; L.orc_fp_pow.argtypes = [_u32, _u64]
        L.orc_rou_fwd.restype = _u32; L.orc_rou_fwd.argtypes = [ctypes.c_uint]
        L.orc_rou_rev.restype = _u32; L.orc_rou_rev.argtypes = [ctypes.c_uint]
        L.orc_circuit_parse.restype = _vp; L.orc_circuit_parse.argtypes = [_vp, _sz]
        L.orc_circuit_free.argtypes = [_vp]
        for f in ("group_size",):
            getattr(L, "orc_circuit_" + f).restype = _u32
            getattr(L, "orc_circuit_" + f).argtypes = [_vp, _u32]
        for f in ("n_taps", "n_global", "n_mix", "n_combos"):
            getattr(L, "orc_circuit_" + f).restype = _u32
            getattr(L, "orc_circuit_" + f).argtypes = [_vp]
        L.orc_prove_segment.restype = _sz
        L.orc_prove_segment.argtypes = [_vp, _vp, _sz, _u32, _vp, _vp, _vp, _vp, _sz]
        L.orc_verify_segment.restype = ctypes.c_int
        L.orc_verify_segment.argtypes = [_vp, _vp, _sz, _vp, _sz]
        L.orc_verify_strerror.restype = ctypes.c_char_p
        L.orc_verify_segment_bound.restype = ctypes.c_int
        L.orc_verify_segment_bound.argtypes = [_vp, _vp, _sz, _vp, _sz, _vp]
        L.orc_code_root.argtypes = [_vp, _u32, _u32, _vp]
        L.orc_witgen.argtypes = [_vp, _u32, _u64, _vp, _vp, _vp]
        L.orc_witgen_public.argtypes = [_vp, _u32, _u64, _vp, _vp, _vp, _vp]
        L.orc_witgen_foreign_code.argtypes = [_vp, _u32, _u64, _u64, _vp, _vp, _vp, _vp]
        L.orc_accum.argtypes = [_vp, _u32, _vp, _vp, _vp, _vp]
        L.orc_accum_public.argtypes = [_vp, _u32, _vp, _vp, _vp, _vp, _vp]
        L.orc_logup_multiplicities.restype = ctypes.c_int
        L.orc_logup_multiplicities.argtypes = [_vp, _u32, _vp, _vp]
        L.orc_logup_totals.argtypes = [_vp, _u32, _vp, _vp, _vp]
        L.orc_circuit_n_late.restype = _u32
        L.orc_circuit_n_late.argtypes = [_vp]
        L.orc_eval_check.argtypes = [_vp, _u32] + [_vp] * 7
        L.orc_poly_ext.argtypes = [_vp] * 6
        L.orc_batch_interpolate_ntt.argtypes = [_vp, _u32, _u32]
        L.orc_batch_expand_into_evaluate_ntt.argtypes = [_vp, _vp, _u32, _u32, _u32]
        L.orc_batch_bit_reverse.argtypes = [_vp, _u32, _u32]
        L.orc_zk_shift.argtypes = [_vp, _u32, _u32]
        L.orc_hash_rows.argtypes = [_vp, _vp, _sz, _sz]
        L.orc_hash_fold.argtypes = [_vp, _sz]
        L.orc_merkle_build.argtypes = [_vp, _vp, _sz, _sz]
        L.orc_hash_elem_slice.argtypes = [_vp, _sz, _vp]
        L.orc_sponge_trace.argtypes = [_vp, _sz, ctypes.c_uint32, _vp]
        L.orc_sponge_trace.restype = ctypes.c_int
        L.orc_hash_pair.argtypes = [_vp, _vp, _vp]
        L.orc_poseidon2_mix.argtypes = [_vp]
        L.orc_poseidon2_consts.argtypes = [_vp, _vp]
        L.orc_batch_evaluate_any.argtypes = [_vp, _u32, _vp, _vp, _u32, _vp]
        L.orc_mix_poly_coeffs.argtypes = [_vp, _vp, _vp, _vp, _vp, _u32, _u32]
        L.orc_eltwise_sum_extelem.argtypes = [_vp, _vp, _u32, _u32]
        L.orc_fri_fold.argtypes = [_vp, _vp, _vp, _u32]
        L.orc_prefix_products.argtypes = [_vp, _u32]
        L.orc_poly_divide.argtypes = [_vp, _u32, _vp, _vp]
        L.orc_poly_interpolate.argtypes = [_vp, _vp, _vp, _u32]
        L.orc_fp4_mul.argtypes = [_vp, _vp, _vp]
        L.orc_fp4_inv.argtypes = [_vp, _vp]
        L.orc_set_threads.argtypes = [ctypes.c_int]
        L.orc_set_threads(min(8, os.cpu_count() or 1))

    # ---- field
    def enc(self, x):
        return self.L.orc_fp_enc(int(x) % P)

    def dec(self, x):
        return self.L.orc_fp_dec(int(x))

    def mul(self, a, b):
        return self.L.orc_fp_mul(int(a), int(b))

    def fp4_mul(self, a, b):
        a, b, o = u32(a), u32(b), np.zeros(4, np.uint32)
        self.L.orc_fp4_mul(_ptr(a), _ptr(b), _ptr(o))
        return o

    def fp4_inv(self, a):
        a, o = u32(a), np.zeros(4, np.uint32)
        self.L.orc_fp4_inv(_ptr(a), _ptr(o))
        return o

    def fp4_pow(self, a, n):
        r = np.array([self.enc(1), 0, 0, 0], np.uint32)
        a = u32(a).copy()
        while n:
            if n & 1:
                r = self.fp4_mul(r, a)
            a = self.fp4_mul(a, a)
            n >>= 1
        return r

    # ---- ops (numpy in, numpy out)
    def batch_interpolate_ntt(self, io, count, po2):
        io = u32(io).copy()
        self.L.orc_batch_interpolate_ntt(_ptr(io), count, po2)
        return io

    def batch_expand_into_evaluate_ntt(self, inp, count, in_po2, expand_bits):
        inp = u32(inp)
        out = np.zeros(count << (in_po2 + expand_bits), np.uint32)
        self.L.orc_batch_expand_into_evaluate_ntt(_ptr(out), _ptr(inp), count, in_po2, expand_bits)
        return out

    def batch_bit_reverse(self, io, count, po2):
        io = u32(io).copy()
        self.L.orc_batch_bit_reverse(_ptr(io), count, po2)
        return io

    def zk_shift(self, io, count, po2):
        io = u32(io).copy()
        self.L.orc_zk_shift(_ptr(io), count, po2)
        return io

    def hash_rows(self, matrix, rows, cols):
        matrix = u32(matrix)
        out = np.zeros(rows * 8, np.uint32)
        self.L.orc_hash_rows(_ptr(out), _ptr(matrix), rows, cols)
        return out

    def hash_fold(self, nodes, output_size):
        nodes = u32(nodes).copy()
        self.L.orc_hash_fold(_ptr(nodes), output_size)
        return nodes

    def merkle_build(self, matrix, rows, cols):
        matrix = u32(matrix)
        nodes = np.zeros(rows * 2 * 8, np.uint32)
        self.L.orc_merkle_build(_ptr(nodes), _ptr(matrix), rows, cols)
        return nodes

    def hash_elem_slice(self, elems):
        elems = u32(elems)
        out = np.zeros(8, np.uint32)
        self.L.orc_hash_elem_slice(_ptr(elems), elems.size, _ptr(out))
        return out

    def sponge_trace(self, words, po2):
        """the rows of the in-circuit sponge over `words` (Montgomery words): [65][2^po2]"""
        words = u32(words)
        out = np.zeros(65 << po2, np.uint32)
        rc = self.L.orc_sponge_trace(_ptr(words), words.size, po2, _ptr(out))
        assert rc == 0, "the words do not fit a trace of 2^%d rows" % po2
        return out.reshape(65, -1)

    def hash_pair(self, a, b):
        a, b, out = u32(a), u32(b), np.zeros(8, np.uint32)
        self.L.orc_hash_pair(_ptr(a), _ptr(b), _ptr(out))
        return out

    def poseidon2_mix(self, cells):
        cells = u32(cells).copy()
        self.L.orc_poseidon2_mix(_ptr(cells))
        return cells

    def poseidon2_consts(self):
        rc, diag = np.zeros(24 * 29, np.uint32), np.zeros(24, np.uint32)
        self.L.orc_poseidon2_consts(_ptr(rc), _ptr(diag))
        return rc, diag

    def batch_evaluate_any(self, coeffs, po2, which, xs):
        coeffs, which, xs = u32(coeffs), u32(which), u32(xs)
        out = np.zeros(4 * which.size, np.uint32)
        self.L.orc_batch_evaluate_any(_ptr(coeffs), po2, _ptr(which), _ptr(xs), which.size, _ptr(out))
        return out

    def mix_poly_coeffs(self, combos, mix_start, mix, inp, combo_of, po2):
        combos, ms, m, inp, co = u32(combos).copy(), u32(mix_start), u32(mix), u32(inp), u32(combo_of)
        self.L.orc_mix_poly_coeffs(_ptr(combos), _ptr(ms), _ptr(m), _ptr(inp), _ptr(co), co.size, po2)
        return combos

    def eltwise_sum_extelem(self, inp, count, n):
        inp = u32(inp)
        out = np.zeros(4 * n, np.uint32)
        self.L.orc_eltwise_sum_extelem(_ptr(out), _ptr(inp), count, n)
        return out

    def fri_fold(self, inp, mix, n_out):
        inp, mix = u32(inp), u32(mix)
        out = np.zeros(4 * n_out, np.uint32)
        self.L.orc_fri_fold(_ptr(out), _ptr(inp), _ptr(mix), n_out)
        return out

    def prefix_products(self, io, n):
        io = u32(io).copy()
        self.L.orc_prefix_products(_ptr(io), n)
        return io

    def poly_divide(self, poly, n, z):
        poly, z, rem = u32(poly).copy(), u32(z), np.zeros(4, np.uint32)
        self.L.orc_poly_divide(_ptr(poly), n, _ptr(z), _ptr(rem))
        return poly, rem

    # ---- circuit
    def circuit(self, blob):
        return OrcCircuit(self, u32(blob))


class OrcCircuit:
    def __init__(self, o, blob):
        self.o, self.blob = o, blob
        self.h = o.L.orc_circuit_parse(_ptr(blob), blob.size)
        assert self.h, "oracle rejected the circuit blob"
        L = o.L
        self.group_size = [L.orc_circuit_group_size(self.h, g) for g in range(3)]
        self.n_taps, self.n_global = L.orc_circuit_n_taps(self.h), L.orc_circuit_n_global(self.h)
        self.n_mix, self.n_combos = L.orc_circuit_n_mix(self.h), L.orc_circuit_n_combos(self.h)
        self.n_late = L.orc_circuit_n_late(self.h)

    def witgen(self, po2, seed, globals_in=None, code_seed=None):
        n = 1 << po2
        code, data = np.zeros(self.group_size[1] * n, np.uint32), np.zeros(self.group_size[2] * n, np.uint32)
        glob = np.zeros(max(self.n_global, 1), np.uint32)
        if code_seed is not None:  # a cheating prover's witness: CODE columns that are not the program's
            gin = None if globals_in is None else u32(globals_in)
            self.o.L.orc_witgen_foreign_code(self.h, po2, seed, code_seed, None if gin is None else _ptr(gin), _ptr(code), _ptr(data), _ptr(glob))
        elif globals_in is not None:
            gin = u32(globals_in)
            assert gin.size == self.n_global
            self.o.L.orc_witgen_public(self.h, po2, seed, _ptr(gin), _ptr(code), _ptr(data), _ptr(glob))
        else:
            self.o.L.orc_witgen(self.h, po2, seed, _ptr(code), _ptr(data), _ptr(glob))
        return code, data, glob[:self.n_global]

    def accum(self, po2, code, data, mix):
        out = np.zeros(self.group_size[0] << po2, np.uint32)
        mix = u32(mix)
        self.o.L.orc_accum(self.h, po2, _ptr(u32(code)), _ptr(u32(data)), _ptr(mix), _ptr(out))
        return out

    def accum_public(self, po2, code, data, glob, mix):
        """the accumulation of a circuit whose argument reads public inputs (the log-derivative argument of the trace circuit)"""
        out = np.zeros(self.group_size[0] << po2, np.uint32)
        self.o.L.orc_accum_public(self.h, po2, _ptr(u32(code)), _ptr(u32(data)), _ptr(u32(glob)), _ptr(u32(mix)), _ptr(out))
        return out

    def logup_multiplicities(self, po2, data, glob):
        """data with the lookup tables' multiplicity columns filled from the lookups its rows make"""
        data = u32(data).copy()
        rc = self.o.L.orc_logup_multiplicities(self.h, po2, _ptr(data), _ptr(u32(glob)))
        assert rc == 0, "a looked-up value is in no table"
        return data

    def logup_totals(self, po2, code, data, glob):
        """glob with the totals of the accumulators that run under public challenges written where the circuit reads them"""
        glob = u32(glob).copy()
        self.o.L.orc_logup_totals(self.h, po2, _ptr(u32(code)), _ptr(u32(data)), _ptr(glob))
        return glob

    def eval_check(self, po2, ea, ec, ed, glob, mix, poly_mix):
        out = np.zeros(16 << po2, np.uint32)
        ea, ec, ed, glob, mix, poly_mix = map(u32, (ea, ec, ed, glob, mix, poly_mix))
        self.o.L.orc_eval_check(self.h, po2, _ptr(ea), _ptr(ec), _ptr(ed), _ptr(glob), _ptr(mix), _ptr(poly_mix), _ptr(out))
        return out

    def prove(self, po2, code, data, glob):
        seal = np.zeros(1 << 21, np.uint32)
        code, data, glob = u32(code), u32(data), u32(glob)
        n = self.o.L.orc_prove_segment(self.h, _ptr(self.blob), self.blob.size, po2, _ptr(code), _ptr(data), _ptr(glob), _ptr(seal), seal.size)
        assert n, "oracle prover failed (non-zero DEEP remainder?)"
        return seal[:n].copy()

    def verify(self, seal, code_root=None):
        seal = u32(seal)
        if code_root is None:
            rc = self.o.L.orc_verify_segment(self.h, _ptr(self.blob), self.blob.size, _ptr(seal), seal.size)
        else:
            root = u32(code_root)
            rc = self.o.L.orc_verify_segment_bound(self.h, _ptr(self.blob), self.blob.size, _ptr(seal), seal.size, _ptr(root))
        return rc, self.o.L.orc_verify_strerror(rc).decode()

    def code_root(self, code, po2):
        """Control root of this circuit's CODE group at 2^po2 rows (what the verifier compares the seal's CODE commitment with)."""
        code = u32(code)
        out = np.zeros(8, dtype=np.uint32)
        self.o.L.orc_code_root(_ptr(code), self.group_size[1], po2, _ptr(out))
        return out


def load():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(os.path.join(ROOT, "oracle", f)) for f in os.listdir(os.path.join(ROOT, "oracle")) if f.endswith((".c", ".h"))):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    return Oracle(ctypes.CDLL(LIB))
