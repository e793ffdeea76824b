"""Host-side harness over libr0hip.so (ctypes), mirroring risc0's `Hal` / `CircuitHal` / segment-prover surface.

The product is the C-ABI library (include/r0hip.h); this module only makes it callable from the tests and bench.py.
The reference's host side is Rust (`risc0_zkvm::default_prover()`, host/src/main.rs:420-423); no Rust toolchain exists
in this image, so the sequencer lives in C++ inside the library and this harness stays a thin binding
(INTEGRATION.md shows the Rust `extern "C"` block a maintainer would add instead).

There is NO CPU fallback: importing works without a GPU (so the ABI can be inspected), but every compute entry point
needs a device and raises R0HipError otherwise.
"""
import ctypes
import sys
import os

import numpy as np

P = 2013265921
INV_RATE = 4
QUERIES = 50
FRI_FOLD = 16
CHECK_SIZE = 16
GROUP_ACCUM, GROUP_CODE, GROUP_DATA = 0, 1, 2

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libr0hip.so")


class R0HipError(RuntimeError):
    pass


_c = ctypes
_vp, _u32, _u64, _sz, _cp = _c.c_void_p, _c.c_uint32, _c.c_uint64, _c.c_size_t, _c.c_char_p
_pp = _c.POINTER(_c.c_void_p)

# name -> argtypes; every entry returns `const char*` (NULL = ok) unless listed in _PLAIN
_SIGNATURES = {
    "r0h_ctx_create": [_c.c_int, _pp],
    "r0h_ctx_destroy": [_vp],
    "r0h_sync": [_vp],
    "r0h_buf_alloc": [_vp, _sz, _pp],
    "r0h_buf_wrap": [_vp, _vp, _sz, _pp],
    "r0h_buf_slice": [_vp, _sz, _sz, _pp],
    "r0h_buf_free": [_vp],
    "r0h_buf_h2d": [_vp, _vp, _sz, _vp, _sz],
    "r0h_buf_d2h": [_vp, _vp, _sz, _vp, _sz],
    "r0h_buf_zero": [_vp, _vp],
    "r0h_batch_interpolate_ntt": [_vp, _vp, _u32, _u32],
    "r0h_batch_expand_into_evaluate_ntt": [_vp, _vp, _vp, _u32, _u32, _u32],
    "r0h_batch_bit_reverse": [_vp, _vp, _u32, _u32],
    "r0h_zk_shift": [_vp, _vp, _u32, _u32],
    "r0h_batch_interpolate_ntt_zk_shift": [_vp, _vp, _u32, _u32],
    "r0h_poseidon2_set_consts": [_vp, _vp, _vp],
    "r0h_hash_rows": [_vp, _vp, _vp, _u32, _u32],
    "r0h_hash_fold": [_vp, _vp, _u32],
    "r0h_merkle_build": [_vp, _vp, _vp, _u32, _u32],
    "r0h_batch_evaluate_any": [_vp, _vp, _u32, _vp, _vp, _u32, _vp],
    "r0h_mix_poly_coeffs": [_vp, _vp, _vp, _vp, _vp, _vp, _u32, _u32],
    "r0h_eltwise_add_elem": [_vp, _vp, _vp, _vp, _u32],
    "r0h_eltwise_copy_elem": [_vp, _vp, _vp, _u32],
    "r0h_eltwise_sum_extelem": [_vp, _vp, _vp, _u32, _u32],
    "r0h_eltwise_zeroize_elem": [_vp, _vp, _u32],
    "r0h_gather_sample": [_vp, _vp, _vp, _u32, _u32, _u32],
    "r0h_scatter": [_vp, _vp, _vp, _vp, _vp, _u32],
    "r0h_fri_fold": [_vp, _vp, _vp, _vp, _u32],
    "r0h_batch_evaluate_any_buf": [_vp, _vp, _u32, _vp, _vp, _u32, _vp],
    "r0h_mix_poly_coeffs_buf": [_vp, _vp, _vp, _vp, _vp, _vp, _u32, _u32],
    "r0h_scatter_slices": [_vp, _vp, _vp, _u32, _vp, _vp, _u32],
    "r0h_hash_fold_io": [_vp, _vp, _u32, _u32],
    "r0h_prefix_products": [_vp, _vp, _u32],
    "r0h_poly_divide": [_vp, _vp, _u32, _vp, _vp],
    "r0h_circuit_emit_hip": [_vp, _sz, _c.POINTER(_c.c_char_p)],
    "r0h_circuit_load": [_vp, _vp, _sz, _cp, _pp],
    "r0h_circuit_free": [_vp],
    "r0h_witgen": [_vp, _vp, _u32, _u64, _vp, _vp, _vp],
    "r0h_witgen_public": [_vp, _vp, _u32, _u64, _vp, _vp, _vp],
    "r0h_seal_digest": [_vp, _sz, _vp],
    "r0h_ctx_set_session_resident_limit": [_vp, _u64],
    "r0h_proof_shrink": [_vp, _vp],
    "r0h_sponge_trace": [_vp, _sz, _u32, _vp],
    "r0h_image_po2": [_vp, _sz, _vp],
    "r0h_image_witness": [_vp, _sz, _u32, _vp, _vp],
    "r0h_prove_image": [_vp, _vp, _vp, _sz, _vp, _vp, _sz, _vp],
    "r0h_accum": [_vp, _vp, _u32, _vp, _vp, _vp, _vp],
    "r0h_eval_check": [_vp, _vp, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "r0h_prove_segment": [_vp, _vp, _u32, _vp, _vp, _vp, _vp, _sz, _c.POINTER(_sz)],
    "r0h_proof_begin": [_vp, _vp, _u32, _vp, _vp, _vp, _vp, _pp],
    "r0h_proof_finish": [_vp, _vp, _vp, _sz, _c.POINTER(_sz)],
    "r0h_proof_abort": [_vp],
    "r0h_verify_seal": [_vp, _sz, _vp, _vp, _vp, _sz, _c.POINTER(_c.c_int), _c.POINTER(_u32)],
    "r0h_verify_seal_bound": [_vp, _sz, _vp, _vp, _vp, _sz, _vp, _c.POINTER(_c.c_int), _c.POINTER(_u32), _vp],
    "r0h_code_root": [_vp, _vp, _u32, _u32, _vp],
    "r0h_code_commit_new": [_vp, _vp, _u32, _u32, _pp],
    "r0h_code_commit_free": [_vp],
    "r0h_code_commit_root": [_vp, _vp],
    "r0h_prove_segment_committed": [_vp, _vp, _u32, _vp, _vp, _vp, _vp, _sz, _c.POINTER(_sz)],
    "r0h_proof_begin_committed": [_vp, _vp, _u32, _vp, _vp, _vp, _vp, _pp],
    "r0h_serde_encode_str": [_vp, _sz, _vp, _sz, _c.POINTER(_sz)],
    "r0h_serde_decode_str": [_vp, _sz, _c.POINTER(_sz), _c.POINTER(_sz), _c.POINTER(_sz)],
    "r0h_journal_commitment_span": [_vp, _sz, _c.POINTER(_sz), _c.POINTER(_sz)],
    "r0h_env_new": [_pp],
    "r0h_env_write_str": [_vp, _vp, _sz],
    "r0h_env_write_u8_seq": [_vp, _vp, _sz],
    "r0h_env_words": [_vp, _pp, _c.POINTER(_sz)],
    "r0h_env_free": [_vp],
    "r0h_receipt_parse": [_vp, _sz, _pp],
    "r0h_receipt_new": [_c.c_int, _vp, _sz, _pp],
    "r0h_receipt_add_segment": [_vp, _vp, _sz, _u32],
    "r0h_receipt_free": [_vp],
    "r0h_receipt_add_segment_claim": [_vp, _vp, _sz, _u32, _vp, _vp],
    "r0h_receipt_segment_claim": [_vp, _sz, _vp, _c.POINTER(_c.c_int)],
    "r0h_receipt_verify": [_vp, _vp, _sz, _vp, _sz, _vp, _c.POINTER(_c.c_int), _c.POINTER(_sz), _c.POINTER(_c.c_int)],
    "r0h_receipt_verify_elf": [_vp, _vp, _sz, _vp, _sz, _vp, _sz, _c.POINTER(_c.c_int), _c.POINTER(_sz), _c.POINTER(_c.c_int)],
    "r0h_receipt_verify_image": [_vp, _vp, _sz, _vp, _sz, _vp, _sz, _vp, _vp, _c.POINTER(_c.c_int), _c.POINTER(_sz), _c.POINTER(_c.c_int)],
    "r0h_receipt_set_image_proof": [_vp, _vp, _sz],
    "r0h_receipt_image_proof": [_vp, _c.POINTER(_vp), _c.POINTER(_sz)],
    "r0h_ctx_set_image_circuit": [_vp, _vp],
    "r0h_sha256": [_vp, _sz, _vp],
    "r0h_image_id_from_hex": [_cp, _vp],
    "r0h_image_id_to_hex": [_vp, _vp],
    "r0h_tagged_struct": [_cp, _vp, _sz, _vp, _sz, _vp],
    "r0h_system_state_digest": [_vp, _vp],
    "r0h_output_digest": [_vp, _sz, _vp, _vp],
    "r0h_claim_digest": [_vp, _vp],
    "r0h_claim_globals": [_vp, _vp],
    "r0h_receipt_journal": [_vp, _pp, _c.POINTER(_sz)],
    "r0h_receipt_segment": [_vp, _sz, _pp, _c.POINTER(_sz), _c.POINTER(_u32)],
    "r0h_receipt_to_json": [_vp, _pp],
    "r0h_ebics_parse": [_vp, _sz, _pp],
    "r0h_ebics_free": [_vp],
    "r0h_ebics_part": [_vp, _c.c_int, _pp, _c.POINTER(_sz)],
    "r0h_ebics_check_digest": [_vp, _c.POINTER(_c.c_int)],
    "r0h_ebics_verify_bank_signature": [_vp, _vp, _sz, _c.POINTER(_c.c_int)],
    "r0h_ebics_check_transaction_key": [_vp, _vp, _sz, _vp, _sz, _vp, _c.POINTER(_c.c_int)],
    "r0h_ebics_verify_witness": [_vp, _vp, _sz, _vp, _sz, _c.POINTER(_c.c_int)],
    "r0h_ebics_decrypt_transaction_key": [_vp, _vp, _sz, _vp, _sz, _c.POINTER(_sz), _vp, _c.POINTER(_c.c_int)],
    "r0h_ebics_witness_sign": [_vp, _vp, _sz, _pp],
    "r0h_ebics_decrypt_order_data": [_vp, _vp],
    "r0h_ebics_document": [_vp, _sz, _pp, _pp, _c.POINTER(_sz)],
    "r0h_rsa_public_key_decimal": [_vp, _sz, _pp, _pp],
    "r0h_ebics_env_inputs": [_vp, _vp, _sz, _vp, _sz, _vp, _sz, _cp, _cp, _vp, _sz, _vp, _sz, _cp, _pp],
    "r0h_aes128_block": [_vp, _vp, _c.c_int, _vp],
    "r0h_zlib_inflate": [_vp, _sz, _pp, _c.POINTER(_sz)],
    "r0h_vm_new": [_pp],
    "r0h_vm_free": [_vp],
    "r0h_vm_load": [_vp, _u32, _vp, _sz],
    "r0h_vm_load_elf": [_vp, _vp, _sz],
    "r0h_vm_set_input": [_vp, _vp, _sz],
    "r0h_vm_set_pc": [_vp, _u32],
    "r0h_vm_set_reg": [_vp, _u32, _u32],
    "r0h_vm_read": [_vp, _u32, _vp, _sz],
    "r0h_vm_run": [_vp, _vp, _c.POINTER(_c.c_int), _c.POINTER(_u32)],
    "r0h_vm_segment_info": [_vp, _sz, _vp],
    "r0h_vm_preflight": [_vp, _sz, _pp, _c.POINTER(_sz)],
    "r0h_vm_trace_witness": [_vp, _sz, _u32, _vp, _vp],
    "r0h_vm_run_segment": [_vp, _vp, _c.POINTER(_c.c_int), _c.POINTER(_c.c_int), _c.POINTER(_u32)],
    "r0h_vm_release_trace": [_vp, _sz],
    "r0h_vm_boundary": [_vp, _sz, _pp, _c.POINTER(_sz)],
    "r0h_trace_witgen": [_vp, _vp, _sz, _vp, _sz, _u32, _vp, _vp, _vp],
    "r0h_logup_multiplicities": [_vp, _vp, _u32, _vp, _vp],
    "r0h_logup_multiplicities_host": [_vp, _sz, _u32, _vp, _vp],
    "r0h_logup_totals": [_vp, _vp, _u32, _vp, _vp, _vp],
    "r0h_accum_public": [_vp, _vp, _u32, _vp, _vp, _vp, _vp, _vp],
    "r0h_prefix_sums": [_vp, _vp, _u32],
    "r0h_proof_data_root": [_vp, _vp],
    "r0h_proof_late": [_vp, _vp, _vp],
    "r0h_proof_globals": [_vp, _vp],
    "r0h_code_commit_columns": [_vp, _pp],
    "r0h_verify_seal_roots": [_vp, _sz, _vp, _sz, _vp, _c.POINTER(_c.c_int), _c.POINTER(_u32), _vp],
    "r0h_session_begin": [_vp, _vp, _vp, _sz, _vp, _sz, _u32, _u64, _u32, _u32, _pp],
    "r0h_session_records": [_vp, _vp, _vp, _sz, _c.POINTER(_sz)],
    "r0h_session_finish": [_vp, _vp, _sz, _pp, _vp, _c.POINTER(_u64)],
    "r0h_session_free": [_vp],
    "r0h_session_challenge": [_vp, _sz, _vp],
    "r0h_last_session_stats": [_vp, _vp],
    "r0h_vm_journal": [_vp, _pp, _c.POINTER(_sz)],
    "r0h_vm_segment_claim": [_vp, _sz, _vp],
    "r0h_compute_image_id": [_vp, _sz, _vp],
    "r0h_camt53_guest_input": [_vp, _vp, _sz, _vp, _sz, _vp, _sz, _vp, _sz, _vp, _sz, _c.c_char_p, _c.c_char_p, _u32, _pp, _c.POINTER(_sz)],
    "r0h_control_root_host": [_vp, _sz, _vp, _vp, _u32, _vp],
    "r0h_prove_elf": [_vp, _vp, _vp, _sz, _vp, _sz, _u32, _u64, _pp, _vp, _c.POINTER(_u64)],
    "r0h_prove_elf_part": [_vp, _vp, _vp, _sz, _vp, _sz, _u32, _u64, _u32, _u32, _pp, _vp, _c.POINTER(_u64)],
    "r0h_receipt_merge": [_pp, _sz, _pp],
    "r0h_recursor_new": [_vp, _vp, _sz, _cp, _u32, _vp, _sz, _vp, _sz, _pp],
    "r0h_recursor_free": [_vp],
    "r0h_recursor_control_root": [_vp, _vp],
    "r0h_lift": [_vp, _vp, _sz, _vp, _pp],
    "r0h_join": [_vp, _vp, _vp, _pp],
    "r0h_node_new": [_vp, _sz, _vp, _pp],
    "r0h_node_free": [_vp],
    "r0h_node_seal": [_vp, _pp, _c.POINTER(_sz)],
    "r0h_node_claim": [_vp, _vp],
    "r0h_node_verify": [_vp, _sz, _vp, _vp, _c.POINTER(_c.c_int)],
    "r0h_kernel_timing": [_vp, _c.c_int],
    "r0h_kernel_stats": [_vp, _vp, _sz],
    "r0h_last_profile": [_vp, _c.POINTER(_c.POINTER(_cp)), _c.POINTER(_c.POINTER(_c.c_float)), _c.POINTER(_u32)],
}
_PLAIN = {
    "r0h_free_error": ([_vp], None),
    "r0h_version": ([], _cp),
    "r0h_receipt_kind": ([_vp], _c.c_int),
    "r0h_receipt_n_segments": ([_vp], _sz),
    "r0h_ebics_n_documents": ([_vp], _sz),
    "r0h_vm_n_segments": ([_vp], _sz),
    "r0h_session_n_segments": ([_vp], _sz),
    "r0h_vm_cycles": ([_vp], _u64),
    "r0h_vm_reg": ([_vp, _u32], _u32),
    "r0h_vm_pc": ([_vp], _u32),
    "r0h_verify_reason": ([_c.c_int], _cp),
    "r0h_trace_column_name": ([_u32], _cp),
    "r0h_receipt_verify_reason": ([_c.c_int], _cp),
    "r0h_buf_device_ptr": ([_vp], _vp),
    "r0h_buf_bytes": ([_vp], _sz),
    "r0h_circuit_group_size": ([_vp, _u32], _u32),
    "r0h_circuit_n_global": ([_vp], _u32),
    "r0h_proof_resident_bytes": ([_vp], _sz),
    "r0h_circuit_n_mix": ([_vp], _u32),
    "r0h_circuit_n_taps": ([_vp], _u32),
}
EXPORTED_SYMBOLS = sorted(list(_SIGNATURES) + list(_PLAIN))

_lib = None


def _torch_runtime_first():
    """A process that uses both this library and torch's GPU side holds two HIP runtimes: libr0hip.so links /opt/rocm's
    libamdhip64.so.7, torch ships its own (no soname: the loader cannot share them).  On the GPU box the one that initialises SECOND
    still finds the device only if it is this library's -- torch's, initialised second, finds none (round 3: gpurun_out/z_sharded2.err).
    So torch's runtime goes first, always: if torch can be imported it is, and its device count is taken, before libr0hip.so is
    opened -- whatever order the caller imports things in.  R0H_NO_TORCH=1 skips this (a process that will never touch torch)."""
    import sys
    if os.environ.get("R0H_NO_TORCH") == "1":
        return
    try:
        if "torch" not in sys.modules:
            import importlib.util
            if importlib.util.find_spec("torch") is None:
                return
        import torch
        if getattr(torch.version, "hip", None):
            torch.cuda.is_available()  # initialises torch's HIP runtime (a device count, nothing more)
    except Exception:  # a broken torch install must not take the library down with it
        pass


def lib():
    """Load libr0hip.so; fails loudly when the HIP library has not been built."""
    global _lib
    if _lib is None:
        _torch_runtime_first()
        if not os.path.exists(LIB_PATH):
            raise R0HipError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950). "
                             "There is no CPU fallback." % LIB_PATH)
        h = ctypes.CDLL(LIB_PATH)
        for name, args in _SIGNATURES.items():
            fn = getattr(h, name)
            fn.argtypes = args
            fn.restype = _vp  # keep the pointer so it can be freed
        for name, (args, res) in _PLAIN.items():
            fn = getattr(h, name)
            fn.argtypes = args
            fn.restype = res
        _lib = h
    return _lib


def _check(err):
    if err:
        msg = ctypes.cast(err, ctypes.c_char_p).value.decode("utf-8", "replace")
        lib().r0h_free_error(err)
        raise R0HipError(msg)


def _u32arr(a):
    a = np.ascontiguousarray(a, dtype=np.uint32)
    return a, a.ctypes.data_as(_vp)


class Buf:
    """A device buffer of uint32 words owned by the library (risc0 `Buffer<Elem>`)."""

    def __init__(self, hal, handle, words):
        self.hal, self.handle, self.words = hal, handle, words

    def to_host(self, offset_words=0, words=None):
        words = self.words - offset_words if words is None else words
        out = np.empty(words, dtype=np.uint32)
        _check(lib().r0h_buf_d2h(self.hal.ctx, self.handle, offset_words * 4, out.ctypes.data_as(_vp), words * 4))
        return out

    def upload(self, array, offset_words=0):
        a, p = _u32arr(array)
        _check(lib().r0h_buf_h2d(self.hal.ctx, self.handle, offset_words * 4, p, a.size * 4))

    def slice(self, offset_words, words):
        h = _vp()
        _check(lib().r0h_buf_slice(self.handle, offset_words * 4, words * 4, ctypes.byref(h)))
        return Buf(self.hal, h, words)

    def device_ptr(self):
        return lib().r0h_buf_device_ptr(self.handle)

    def free(self):
        if self.handle:
            _check(lib().r0h_buf_free(self.handle))
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Circuit:
    def __init__(self, hal, handle, blob):
        self.hal, self.handle, self.blob = hal, handle, blob
        L = lib()
        self.group_size = [L.r0h_circuit_group_size(handle, g) for g in range(3)]
        self.n_global = L.r0h_circuit_n_global(handle)
        self.n_mix = L.r0h_circuit_n_mix(handle)
        self.n_taps = L.r0h_circuit_n_taps(handle)

    def free(self):
        if self.handle:
            _check(lib().r0h_circuit_free(self.handle))
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class CodeCommit:
    """The CODE group of a circuit committed once for one trace size (r0h_code_commit_new): every proof of that size on any
    context of the same device reads it instead of committing CODE again."""

    def __init__(self, handle, po2):
        self.handle, self.po2 = handle, po2

    def root(self):
        out = np.zeros(8, dtype=np.uint32)
        _check(lib().r0h_code_commit_root(self.handle, out.ctypes.data_as(_vp)))
        return out

    def free(self):
        if self.handle:
            _check(lib().r0h_code_commit_free(self.handle))
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def emit_eval_check_source(blob):
    """HIP source of the circuit's eval_check kernels (pure host; needs no GPU)."""
    a, p = _u32arr(blob)
    out = ctypes.c_char_p()
    _check(lib().r0h_circuit_emit_hip(p, a.size, ctypes.byref(out)))
    src = out.value.decode()
    lib().r0h_free_error(ctypes.cast(out, _vp))
    return src


def verify_seal(blob, seal, poseidon2_consts=None, code_root=None):
    """Host-side check of a seal against a circuit blob (r0h_verify_seal_bound; needs no GPU): returns (verdict, reason, po2),
    verdict 0 = accepted.  poseidon2_consts = (round_constants[29*24], diag_m1[24]) canonical words, or None for the
    compiled-in table.  code_root = the program's control root at the seal's trace size (Hal.code_root); None leaves the
    seal unbound to any program (proof-system tests only).  Mirrors `receipt.verify(image_id)` (verifier/src/main.rs:124-126)."""
    b, pb = _u32arr(blob)
    s_, ps = _u32arr(seal)
    prc = pdg = None
    if poseidon2_consts is not None:
        rc, prc = _u32arr(poseidon2_consts[0])
        dg, pdg = _u32arr(poseidon2_consts[1])
        if rc.size != 29 * 24 or dg.size != 24:
            raise R0HipError("verify_seal: Poseidon2 tables must hold 29*24 and 24 words")
    verdict, po2 = _c.c_int(-1), _u32(0)
    proot = None
    if code_root is not None:
        root, proot = _u32arr(code_root)
        if root.size != 8:
            raise R0HipError("verify_seal: a code root is 8 words")
    _check(lib().r0h_verify_seal_bound(pb, b.size, prc, pdg, ps, s_.size, proot, ctypes.byref(verdict), ctypes.byref(po2), None))
    return verdict.value, lib().r0h_verify_reason(verdict.value).decode(), po2.value


def seal_code_root(blob, seal):
    """The CODE root a seal commits to (read out while verifying it; zeros if the seal is rejected before that point)."""
    b, pb = _u32arr(blob)
    s_, ps = _u32arr(seal)
    verdict, po2 = _c.c_int(-1), _u32(0)
    out = np.zeros(8, dtype=np.uint32)
    _check(lib().r0h_verify_seal_bound(pb, b.size, None, None, ps, s_.size, None, ctypes.byref(verdict), ctypes.byref(po2), out.ctypes.data_as(_vp)))
    return out


def serde_encode_str(text):
    """risc0 serde word stream of a String ([u32 LE length][utf8][zero padding to 4]): ExecutorEnv inputs and journal.bytes."""
    raw = text.encode("utf-8") if isinstance(text, str) else bytes(text)
    n = _sz(0)
    _check(lib().r0h_serde_encode_str(raw, len(raw), None, 0, ctypes.byref(n)))
    out = (ctypes.c_uint8 * n.value)()
    _check(lib().r0h_serde_encode_str(raw, len(raw), out, n.value, ctypes.byref(n)))
    return bytes(out)


def serde_decode_str(data):
    """Inverse of serde_encode_str: returns (bytes of the string, bytes consumed); raises R0HipError on bad framing."""
    data = bytes(data)
    off, ln, used = _sz(0), _sz(0), _sz(0)
    _check(lib().r0h_serde_decode_str(data, len(data), ctypes.byref(off), ctypes.byref(ln), ctypes.byref(used)))
    return data[off.value:off.value + ln.value], used.value


def env_input_words(items):
    """The u32 words `ExecutorEnv::builder().write(&x)...` produces for a list of inputs (host/src/main.rs:389-417):
    str -> String frame, bytes -> Vec<u8> (one word per byte)."""
    h = _vp()
    _check(lib().r0h_env_new(ctypes.byref(h)))
    try:
        for it in items:
            if isinstance(it, str):
                raw = it.encode("utf-8")
                _check(lib().r0h_env_write_str(h, raw, len(raw)))
            else:
                raw = bytes(it)
                _check(lib().r0h_env_write_u8_seq(h, raw, len(raw)))
        p, n = _vp(), _sz(0)
        _check(lib().r0h_env_words(h, ctypes.byref(p), ctypes.byref(n)))
        return np.frombuffer(ctypes.string_at(p, n.value * 4), dtype=np.uint32).copy()
    finally:
        lib().r0h_env_free(h)


def journal_commitment(data):
    """The commitment JSON text hyperfridge reads out of journal.bytes: first '{' .. last '}' (host/src/main.rs:258-267)."""
    data = bytes(data)
    off, ln = _sz(0), _sz(0)
    _check(lib().r0h_journal_commitment_span(data, len(data), ctypes.byref(off), ctypes.byref(ln)))
    return data[off.value:off.value + ln.value]


def seal_digest(seal):
    """8-word Poseidon2 name of a seal (r0h_seal_digest): what a recursion step's public inputs carry."""
    a, pa = _u32arr(seal)
    out = np.zeros(8, dtype=np.uint32)
    _check(lib().r0h_seal_digest(pa, a.size, out.ctypes.data_as(_vp)))
    return out


SPONGE_DATA_COLUMNS = 65
IMAGE_COLUMNS, IMAGE_GLOBALS, IMAGE_GAMMA, IMAGE_SUM = 69, 28, 8, 24  # R0H_IMAGE_*


def image_po2(elf):
    """smallest trace of the image circuit that holds the sponge over this ELF's words (r0h_image_po2)"""
    buf = (ctypes.c_uint8 * len(elf)).from_buffer_copy(elf)
    out = _u32(0)
    _check(lib().r0h_image_po2(buf, len(elf), ctypes.byref(out)))
    return out.value


def image_witness(elf, po2=None):
    """The image circuit's DATA group for an ELF and its public inputs with the digest filled in (r0h_image_witness):
    ([IMAGE_COLUMNS][2^po2] Montgomery words, IMAGE_GLOBALS words)."""
    po2 = image_po2(elf) if po2 is None else po2
    buf = (ctypes.c_uint8 * len(elf)).from_buffer_copy(elf)
    data, glob = np.zeros((IMAGE_COLUMNS, 1 << po2), dtype=np.uint32), np.zeros(IMAGE_GLOBALS, dtype=np.uint32)
    _check(lib().r0h_image_witness(buf, len(elf), po2, data.ctypes.data_as(_vp), glob.ctypes.data_as(_vp)))
    return data, glob


def sponge_trace(words, po2):
    """The rows of the recursion circuit's in-circuit Poseidon2 sponge over `words` (r0h_sponge_trace): [65][2^po2] -- st[24], aux[24],
    in[16], act.  What r0h_lift / r0h_join plant into a node's witness."""
    a, pa = _u32arr(words)
    out = np.zeros((SPONGE_DATA_COLUMNS, 1 << po2), dtype=np.uint32)
    _check(lib().r0h_sponge_trace(pa, a.size, po2, out.ctypes.data_as(_vp)))
    return out


class Ebics:
    """An EBICS response pre-processed as data/checkResponse.sh does it (r0h_ebics_*; host only, no GPU)."""
    AUTHENTICATED, SIGNED_INFO, SIGNATURE_VALUE, ORDER_DATA, DIGEST_VALUE, SIGNATURE_BIN, TRANSACTION_KEY_BIN, ORDER_DATA_BIN, PAYLOAD_ZIP = range(9)

    def __init__(self, xml):
        raw = xml.encode("utf-8") if isinstance(xml, str) else bytes(xml)
        self.handle = _vp()
        _check(lib().r0h_ebics_parse(raw, len(raw), ctypes.byref(self.handle)))

    def part(self, which):
        p, n = _vp(), _sz(0)
        _check(lib().r0h_ebics_part(self.handle, which, ctypes.byref(p), ctypes.byref(n)))
        return ctypes.string_at(p, n.value) if n.value else b""

    def camt53_guest_input(self, pub_bank_pem, pub_client_pem, pub_witness_pem, tx_key_block, witness_hex, iban, host_info, form=1):
        """r0h_camt53_guest_input: the input word stream of this library's camt53 guest (tools/guest_camt53.py input_stream) -> uint32 array"""
        pems = [p.encode() if isinstance(p, str) else bytes(p) for p in (pub_bank_pem, pub_client_pem, pub_witness_pem)]
        block = bytes(tx_key_block)
        hx = witness_hex.encode() if isinstance(witness_hex, str) else bytes(witness_hex)
        words, n = _vp(), _sz(0)
        _check(lib().r0h_camt53_guest_input(self.handle, pems[0], len(pems[0]), pems[1], len(pems[1]), pems[2], len(pems[2]), block, len(block), hx, len(hx),
                                            iban.encode(), host_info.encode(), form, ctypes.byref(words), ctypes.byref(n)))
        try:
            return np.frombuffer(ctypes.string_at(words, 4 * n.value), dtype=np.uint32).copy()
        finally:
            lib().r0h_free_error(words)

    def _flag(self, fn, *args):
        ok = _c.c_int(-1)
        _check(fn(self.handle, *args, ctypes.byref(ok)))
        return bool(ok.value)

    def check_digest(self):
        return self._flag(lib().r0h_ebics_check_digest)

    def verify_bank_signature(self, pub_bank_pem):
        pem = pub_bank_pem.encode() if isinstance(pub_bank_pem, str) else bytes(pub_bank_pem)
        return self._flag(lib().r0h_ebics_verify_bank_signature, pem, len(pem))

    def check_transaction_key(self, pub_client_pem, raw_block):
        """(ok, 16-byte AES key) for the raw RSA-decrypted block 00 02 PS 00 key."""
        pem = pub_client_pem.encode() if isinstance(pub_client_pem, str) else bytes(pub_client_pem)
        raw, key = bytes(raw_block), (ctypes.c_uint8 * 16)()
        ok = self._flag(lib().r0h_ebics_check_transaction_key, pem, len(pem), raw, len(raw), key)
        return ok, bytes(key)

    def decrypt_transaction_key(self, client_private_pem):
        """(ok, raw RSA-decrypted block, 16-byte AES key) with the client's PRIVATE key (checkResponse.sh:231-236)."""
        pem = client_private_pem.encode() if isinstance(client_private_pem, str) else bytes(client_private_pem)
        raw, n, key, ok = (ctypes.c_uint8 * 1024)(), _sz(0), (ctypes.c_uint8 * 16)(), _c.c_int(-1)
        _check(lib().r0h_ebics_decrypt_transaction_key(self.handle, pem, len(pem), raw, 1024, ctypes.byref(n), key, ctypes.byref(ok)))
        return bool(ok.value), bytes(raw[:n.value]), bytes(key)

    def witness_sign(self, witness_private_pem):
        """The witness signature over SHA-256(decoded order data) as the `xxd -p` text of `<xml>-Witness.hex` (checkResponse.sh:276-279)."""
        pem = witness_private_pem.encode() if isinstance(witness_private_pem, str) else bytes(witness_private_pem)
        p = _vp()
        _check(lib().r0h_ebics_witness_sign(self.handle, pem, len(pem), ctypes.byref(p)))
        text = ctypes.cast(p, _cp).value
        lib().r0h_free_error(p)
        return text

    def verify_witness(self, pub_witness_pem, witness_hex):
        pem = pub_witness_pem.encode() if isinstance(pub_witness_pem, str) else bytes(pub_witness_pem)
        hx = witness_hex.encode() if isinstance(witness_hex, str) else bytes(witness_hex)
        return self._flag(lib().r0h_ebics_verify_witness, pem, len(pem), hx, len(hx))

    def decrypt_order_data(self, key):
        """AES-128-CBC (zero IV) -> inflate -> ZIP: returns [(member name, bytes)]."""
        _check(lib().r0h_ebics_decrypt_order_data(self.handle, bytes(key)))
        docs = []
        for i in range(lib().r0h_ebics_n_documents(self.handle)):
            name, data, n = _vp(), _vp(), _sz(0)
            _check(lib().r0h_ebics_document(self.handle, i, ctypes.byref(name), ctypes.byref(data), ctypes.byref(n)))
            docs.append((ctypes.cast(name, _cp).value.decode("utf-8", "replace"), ctypes.string_at(data, n.value)))
        return docs

    def env_inputs(self, pub_bank_pem, client_private_pem, decrypted_tx_key, iban, host_info, witness_hex, pub_witness_pem, verbose):
        """The u32 words of the thirteen ExecutorEnv inputs (host/src/main.rs:389-417)."""
        b = lambda x: x.encode("utf-8") if isinstance(x, str) else bytes(x)
        bank, client, wit, pubw, tx = b(pub_bank_pem), b(client_private_pem), b(witness_hex), b(pub_witness_pem), bytes(decrypted_tx_key)
        env = _vp()
        _check(lib().r0h_ebics_env_inputs(self.handle, bank, len(bank), client, len(client), tx, len(tx), b(iban), b(host_info), wit, len(wit), pubw, len(pubw),
                                          b(verbose), ctypes.byref(env)))
        try:
            p, n = _vp(), _sz(0)
            _check(lib().r0h_env_words(env, ctypes.byref(p), ctypes.byref(n)))
            return np.frombuffer(ctypes.string_at(p, n.value * 4), dtype=np.uint32).copy()
        finally:
            lib().r0h_env_free(env)

    def close(self):
        if self.handle:
            lib().r0h_ebics_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def rsa_public_key_decimal(pem):
    """(modulus, exponent) of a PEM "PUBLIC KEY" as decimal strings, as host/src/main.rs:383-387 hands the bank key to the guest."""
    raw = pem.encode() if isinstance(pem, str) else bytes(pem)
    m, e = _vp(), _vp()
    _check(lib().r0h_rsa_public_key_decimal(raw, len(raw), ctypes.byref(m), ctypes.byref(e)))
    out = ctypes.cast(m, _cp).value.decode(), ctypes.cast(e, _cp).value.decode()
    lib().r0h_free_error(m)
    lib().r0h_free_error(e)
    return out


def aes128_block(key, block, decrypt=False):
    out = (ctypes.c_uint8 * 16)()
    _check(lib().r0h_aes128_block(bytes(key), bytes(block), 1 if decrypt else 0, out))
    return bytes(out)


def zlib_inflate(data):
    data = bytes(data)
    p, n = _vp(), _sz(0)
    _check(lib().r0h_zlib_inflate(data, len(data), ctypes.byref(p), ctypes.byref(n)))
    out = ctypes.string_at(p, n.value)
    lib().r0h_free_error(p)
    return out


class SystemState(ctypes.Structure):
    """risc0-binfmt `SystemState` (r0h_system_state)."""
    _fields_ = [("pc", _u32), ("merkle_root", ctypes.c_uint8 * 32)]

    @classmethod
    def make(cls, pc, merkle_root):
        st = cls()
        st.pc = pc
        st.merkle_root[:] = bytes(merkle_root)
        return st

    def digest(self):
        out = (ctypes.c_uint8 * 32)()
        _check(lib().r0h_system_state_digest(ctypes.byref(self), out))
        return bytes(out)


class ReceiptClaim(ctypes.Structure):
    """risc0-zkvm `ReceiptClaim`, flattened (r0h_receipt_claim): exit code as (system, user), input/output as digests (zeros = None)."""
    _fields_ = [("pre", SystemState), ("post", SystemState), ("exit_system", _u32), ("exit_user", _u32),
                ("input_digest", ctypes.c_uint8 * 32), ("output_digest", ctypes.c_uint8 * 32)]
    HALTED, PAUSED, SPLIT = 0, 1, 2

    @classmethod
    def make(cls, pre, post, exit_system, exit_user=0, output_digest=None, input_digest=None):
        c = cls()
        c.pre, c.post, c.exit_system, c.exit_user = pre, post, exit_system, exit_user
        if output_digest is not None:
            c.output_digest[:] = bytes(output_digest)
        if input_digest is not None:
            c.input_digest[:] = bytes(input_digest)
        return c

    def digest(self):
        out = (ctypes.c_uint8 * 32)()
        _check(lib().r0h_claim_digest(ctypes.byref(self), out))
        return bytes(out)

    def globals(self):
        """The eight public-input words that name this claim in a seal (r0h_claim_globals)."""
        return claim_globals(self.digest())


TRACE_COLUMNS = 128  # R0H_TRACE_COLUMNS
# R0H_TRACE_GLOBALS: claim words 0..7, first pc, pc after the last cycle, cycles, end kind (0 cut / 1 HALT / 2 PAUSE), kind != 0, exit code halves, segment
# number, closing, number x (1 - closing), first / last boundary address; LATE (the last 20): the session challenge (16), the segment's sum under it (4)
TRACE_GLOBALS = 40
TRACE_LATE_GLOBALS = 20
TRACE_GAMMA, TRACE_SUM = 20, 36
SESSION_RECORD_WORDS = 28
TRACE_MIN_PO2, TRACE_MAX_PO2 = 16, 21
JOURNAL_BASE = 0x20000000  # R0H_JOURNAL_BASE: journal word i is the word at JOURNAL_BASE + 4 i when COMMIT names it
REG_BASE = 0x10000000  # R0H_REG_BASE: address of x[i] in the trace circuit's one address space (memory: word index)
MEM_NONE, MEM_READ, MEM_WRITE = 0, 1, 2  # r0h_preflight_row.mem_kind


def trace_column_names():
    names, k = [], 0
    while True:
        s = lib().r0h_trace_column_name(k)
        if s is None:
            return names
        names.append(s.decode())
        k += 1


class VmLimits(ctypes.Structure):
    _fields_ = [("segment_po2", _u32), ("page_in_cycles", _u32), ("page_out_cycles", _u32), ("keep_trace", _u32), ("max_cycles", _u64),
                ("boundary_rows", _u32), ("reserved", _u32)]


class VmSegment(ctypes.Structure):
    _fields_ = [("index", _u32), ("exit_system", _u32), ("exit_user", _u32), ("pages_in", _u32), ("pages_out", _u32), ("boundary_rows", _u32),
                ("closing", _u32), ("reserved", _u32), ("user_cycles", _u64), ("paging_cycles", _u64), ("pre", SystemState), ("post", SystemState)]


class PreflightRow(ctypes.Structure):
    _fields_ = [("cycle", _u32), ("pc", _u32), ("insn", _u32), ("next_pc", _u32), ("rs1_value", _u32), ("rs2_value", _u32), ("rd", _u32), ("rd_before", _u32),
                ("rd_after", _u32), ("mem_kind", _u32), ("mem_addr", _u32), ("mem_before", _u32), ("mem_after", _u32), ("prev", _u32 * 5)]


class SessionStats(ctypes.Structure):
    _fields_ = [("segments", _u32), ("lean_segments", _u32), ("cycles", _u64), ("executor_s", _c.c_double), ("witgen_ms", _c.c_double), ("prove_ms", _c.c_double), ("wall_s", _c.c_double)]


class PreflightBound(ctypes.Structure):
    _fields_ = [("addr", _u32), ("first_value", _u32), ("last_value", _u32), ("last_ts", _u32), ("prev_seg", _u32), ("init_value", _u32), ("flags", _u32), ("reserved", _u32)]


class TraceSegment(ctypes.Structure):
    _fields_ = [("number", _u32), ("closing", _u32), ("idle_pc", _u32), ("reserved", _u32)]


class Vm:
    """RV32IM executor + segmenter + preflight trace (r0h_vm_*; host only): the step of `prover.prove(env, elf)` before prove_segment."""
    HALTED, PAUSED, LIMIT = 0, 1, 2

    def __init__(self):
        self.handle = _vp()
        _check(lib().r0h_vm_new(ctypes.byref(self.handle)))

    def load(self, addr, words):
        a, pa = _u32arr(words)
        _check(lib().r0h_vm_load(self.handle, addr, pa, a.size))

    def load_elf(self, data):
        data = bytes(data)
        _check(lib().r0h_vm_load_elf(self.handle, data, len(data)))

    def set_input(self, words):
        a, pa = _u32arr(words)
        _check(lib().r0h_vm_set_input(self.handle, pa, a.size))

    def set_pc(self, pc):
        _check(lib().r0h_vm_set_pc(self.handle, pc))

    def set_reg(self, i, v):
        _check(lib().r0h_vm_set_reg(self.handle, i, v & 0xFFFFFFFF))

    def reg(self, i):
        return lib().r0h_vm_reg(self.handle, i)

    @property
    def pc(self):
        return lib().r0h_vm_pc(self.handle)

    @property
    def cycles(self):
        return lib().r0h_vm_cycles(self.handle)

    def read(self, addr, n):
        out = np.zeros(n, dtype=np.uint32)
        _check(lib().r0h_vm_read(self.handle, addr, out.ctypes.data_as(_vp), n))
        return out

    def run(self, segment_po2=20, page_in_cycles=0, page_out_cycles=0, keep_trace=False, max_cycles=0, boundary_rows=False):
        """Returns (exit kind, exit code); raises R0HipError on a guest trap."""
        lim = VmLimits(segment_po2, page_in_cycles, page_out_cycles, 1 if keep_trace else 0, max_cycles, 1 if boundary_rows else 0, 0)
        kind, code = _c.c_int(-1), _u32(0)
        _check(lib().r0h_vm_run(self.handle, ctypes.byref(lim), ctypes.byref(kind), ctypes.byref(code)))
        return kind.value, code.value

    def run_segment(self, segment_po2=20, page_in_cycles=0, page_out_cycles=0, keep_trace=False, max_cycles=0, boundary_rows=False):
        """One more segment of the run (r0h_vm_run_segment): returns (finished, exit kind, exit code)."""
        lim = VmLimits(segment_po2, page_in_cycles, page_out_cycles, 1 if keep_trace else 0, max_cycles, 1 if boundary_rows else 0, 0)
        fin, kind, code = _c.c_int(0), _c.c_int(-1), _u32(0)
        _check(lib().r0h_vm_run_segment(self.handle, ctypes.byref(lim), ctypes.byref(fin), ctypes.byref(kind), ctypes.byref(code)))
        return bool(fin.value), kind.value, code.value

    def release_trace(self, i):
        _check(lib().r0h_vm_release_trace(self.handle, i))

    def segments(self):
        out = []
        for i in range(lib().r0h_vm_n_segments(self.handle)):
            s = VmSegment()
            _check(lib().r0h_vm_segment_info(self.handle, i, ctypes.byref(s)))
            out.append(s)
        return out

    def trace_witness(self, i, po2, claim_globals=None, multiplicities=True):
        """DATA group of the trace circuit (TRACE_COLUMNS x 2^po2, Montgomery words, column-major) from segment i's preflight and
        boundary rows on the HOST (r0h_vm_trace_witness: the reference the device kernel is compared with), and its TRACE_GLOBALS
        public inputs: claim_globals (8 words, zeros when None), first pc, pc after the last cycle, cycles, how it ends, exit code."""
        data = np.zeros(TRACE_COLUMNS << po2, dtype=np.uint32)
        glob = np.zeros(TRACE_GLOBALS, dtype=np.uint32)
        _check(lib().r0h_vm_trace_witness(self.handle, i, po2, data.ctypes.data_as(_vp), glob.ctypes.data_as(_vp)))
        if claim_globals is not None:
            glob[:8] = claim_globals
        if multiplicities:  # the lookup tables' multiplicity columns, from the circuit's own description of its lookups
            blob = trace_blob()
            _check(lib().r0h_logup_multiplicities_host(blob.ctypes.data_as(_vp), blob.size, po2, data.ctypes.data_as(_vp), glob.ctypes.data_as(_vp)))
        return data, glob

    def _preflight_raw(self, i):
        p, n = _vp(), _sz(0)
        _check(lib().r0h_vm_preflight(self.handle, i, ctypes.byref(p), ctypes.byref(n)))
        q, m = _vp(), _sz(0)
        _check(lib().r0h_vm_boundary(self.handle, i, ctypes.byref(q), ctypes.byref(m)))
        return p, n.value, q, m.value

    def preflight(self, i):
        p, n, _, _ = self._preflight_raw(i)
        rows = ctypes.cast(p, ctypes.POINTER(PreflightRow))
        return [rows[k] for k in range(n)]

    def boundary(self, i):
        _, _, q, m = self._preflight_raw(i)
        rows = ctypes.cast(q, ctypes.POINTER(PreflightBound))
        return [rows[k] for k in range(m)]

    def preflight_arrays(self, i):
        """The compact rows of segment i as numpy arrays (copies): [n, 18] uint32 cycles, [m, 8] uint32 boundary rows."""
        p, n, q, m = self._preflight_raw(i)
        rows = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(_u32)), shape=(n, 18)).copy() if n else np.zeros((0, 18), np.uint32)
        bounds = np.ctypeslib.as_array(ctypes.cast(q, ctypes.POINTER(_u32)), shape=(m, 8)).copy() if m else np.zeros((0, 8), np.uint32)
        return rows, bounds

    @property
    def journal(self):
        p, n = _vp(), _sz(0)
        _check(lib().r0h_vm_journal(self.handle, ctypes.byref(p), ctypes.byref(n)))
        return ctypes.string_at(p, n.value) if n.value else b""

    def claims(self):
        out = []
        for i in range(lib().r0h_vm_n_segments(self.handle)):
            c = ReceiptClaim()
            _check(lib().r0h_vm_segment_claim(self.handle, i, ctypes.byref(c)))
            out.append(c)
        return out

    def close(self):
        if self.handle:
            lib().r0h_vm_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def session_challenge(records):
    """the 16 words all seals of a trace-circuit session carry at TRACE_GAMMA (r0h_session_challenge): records [n, SESSION_RECORD_WORDS]
    = every segment's early public inputs and DATA root, in index order"""
    rec = np.ascontiguousarray(records, dtype=np.uint32).reshape(-1, SESSION_RECORD_WORDS)
    out = np.zeros(16, dtype=np.uint32)
    _check(lib().r0h_session_challenge(rec.ctypes.data_as(_vp), rec.shape[0], out.ctypes.data_as(_vp)))
    return out


_trace_blob = None


def trace_blob():
    """circuits/trace.r0c (built by __graft_entry__.build())"""
    global _trace_blob
    if _trace_blob is None:
        _trace_blob = np.fromfile(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "circuits", "trace.r0c"), dtype=np.uint32)
    return _trace_blob


def control_root_host(blob, po2, round_constants=None, diag_m1=None):
    """r0h_control_root_host: the control root of the circuit's own CODE columns at trace size 2^po2, computed on the host from the blob
    (the 8 words Hal.code_commit(...).root() gives on the device)"""
    blob, pb = _u32arr(blob)
    out = np.zeros(8, dtype=np.uint32)
    rc = dg = None
    prc = pdg = None
    if round_constants is not None:
        rc, prc = _u32arr(round_constants)
        dg, pdg = _u32arr(diag_m1)
    _check(lib().r0h_control_root_host(pb, blob.size, prc, pdg, po2, out.ctypes.data_as(_vp)))
    return out


def compute_image_id(elf):
    """r0h_compute_image_id: the 32-byte image id of an ELF (risc0-binfmt `compute_image_id`; what prove_elf returns beside the receipt)"""
    elf = bytes(elf)
    out = (ctypes.c_uint8 * 32)()
    _check(lib().r0h_compute_image_id(elf, len(elf), out))
    return bytes(out)


def image_id_from_hex(text):
    """32 digest bytes from the reference's image-id text (host/out/IMAGE_ID.hex: eight {:08x} u32 words, little-endian in the digest)."""
    out = (ctypes.c_uint8 * 32)()
    _check(lib().r0h_image_id_from_hex(text.encode() if isinstance(text, str) else bytes(text), out))
    return bytes(out)


def image_id_to_hex(image_id):
    out = ctypes.create_string_buffer(65)
    _check(lib().r0h_image_id_to_hex(bytes(image_id), out))
    return out.value.decode()


def sha256(data):
    data = bytes(data)
    out = (ctypes.c_uint8 * 32)()
    _check(lib().r0h_sha256(data, len(data), out))
    return bytes(out)


def tagged_struct(tag, down, data):
    """risc0-binfmt `tagged_struct`: SHA-256(SHA-256(tag) || down digests || data as u32 LE || len(down) as u16 LE)."""
    flat = b"".join(bytes(d) for d in down)
    arr, parr = _u32arr(list(data) or [0])
    out = (ctypes.c_uint8 * 32)()
    _check(lib().r0h_tagged_struct(tag.encode(), flat, len(down), parr, len(data), out))
    return bytes(out)


def output_digest(journal, assumptions_digest=None):
    journal = bytes(journal)
    out = (ctypes.c_uint8 * 32)()
    _check(lib().r0h_output_digest(journal, len(journal), assumptions_digest, out))
    return bytes(out)


def claim_globals(claim_digest):
    out = np.zeros(8, dtype=np.uint32)
    _check(lib().r0h_claim_globals(bytes(claim_digest), out.ctypes.data_as(_vp)))
    return out


def session_claims(n_segments, journal, state_seed=b"r0hip synthetic session"):
    """The claims of an n-segment session with synthetic system states (no executor exists here: the states are SHA-256 names,
    not memory images): segment k runs from state k to state k+1, all but the last end in SystemSplit, the last halts with the
    journal's output.  Returns (claims, image_id) with image_id = digest of state 0."""
    states = [SystemState.make(0, sha256(state_seed + b"/" + str(k).encode())) for k in range(n_segments + 1)]
    claims = []
    for k in range(n_segments):
        last = k == n_segments - 1
        claims.append(ReceiptClaim.make(states[k], states[k + 1], ReceiptClaim.HALTED if last else ReceiptClaim.SPLIT, 0,
                                        output_digest(journal) if last else None))
    return claims, states[0].digest()


class Receipt:
    """Receipt JSON envelope (host/src/main.rs:251-252 writes it, verifier/src/main.rs:118-119 reads it)."""

    def __init__(self, handle):
        self.handle = handle

    @classmethod
    def merge(cls, receipts):
        """r0h_receipt_merge: composite receipts that each hold some segments of one session -> the receipt of the session"""
        arr = (_vp * len(receipts))(*[r.handle for r in receipts])
        h = _vp()
        _check(lib().r0h_receipt_merge(arr, len(receipts), ctypes.byref(h)))
        return cls(h)

    @classmethod
    def parse(cls, text):
        raw = text.encode("utf-8") if isinstance(text, str) else bytes(text)
        h = _vp()
        _check(lib().r0h_receipt_parse(raw, len(raw), ctypes.byref(h)))
        return cls(h)

    @classmethod
    def new(cls, journal, seals=None, claims=None, indices=None):
        """a Fake receipt (seals None) or a composite one of the given seals (and claims); indices: the segments' indices when the
        receipt holds only some segments of a session (a rank's share: Receipt.merge)"""
        journal = bytes(journal)
        h = _vp()
        _check(lib().r0h_receipt_new(0 if seals is None else 1, journal, len(journal), ctypes.byref(h)))
        rc = cls(h)
        for k, seal in enumerate(seals or []):
            a, pa = _u32arr(seal)
            i = k if indices is None else indices[k]
            if claims is None:
                _check(lib().r0h_receipt_add_segment(h, pa, a.size, i))
            else:
                _check(lib().r0h_receipt_add_segment_claim(h, pa, a.size, i, ctypes.byref(claims[k]), None))
        return rc

    def claims(self):
        out = []
        for i in range(lib().r0h_receipt_n_segments(self.handle)):
            c, has = ReceiptClaim(), _c.c_int(0)
            _check(lib().r0h_receipt_segment_claim(self.handle, i, ctypes.byref(c), ctypes.byref(has)))
            out.append(c if has.value else None)
        return out

    def verify(self, blob, control_roots, image_id, elf=None):
        """`receipt.verify(image_id)` (r0h_receipt_verify): control_roots = {po2: root[8]}; image_id is required, as it is in the
        reference (None is passed through and yields verdict 12 "not tied to a program", never 0).  A receipt over the trace circuit
        binds its program through the session sum: give the ELF (r0h_receipt_verify_elf: its image id is derived, image_id is then
        ignored); with an image id alone the verdict is at best 15 "the program image was not given".  Returns (verdict, reason,
        segment, seal verdict)."""
        b, pb = _u32arr(blob)
        table = np.zeros(9 * max(len(control_roots), 1), dtype=np.uint32)
        for k, (po2, root) in enumerate(sorted(control_roots.items())):
            table[9 * k] = po2
            table[9 * k + 1:9 * k + 9] = root
        verdict, seg, sv = _c.c_int(-1), _sz(0), _c.c_int(0)
        if elf is not None:
            elf = bytes(elf)
            _check(lib().r0h_receipt_verify_elf(self.handle, pb, b.size, table.ctypes.data_as(_vp), len(control_roots), elf, len(elf),
                                                ctypes.byref(verdict), ctypes.byref(seg), ctypes.byref(sv)))
        else:
            _check(lib().r0h_receipt_verify(self.handle, pb, b.size, table.ctypes.data_as(_vp), len(control_roots), None if image_id is None else bytes(image_id),
                                            ctypes.byref(verdict), ctypes.byref(seg), ctypes.byref(sv)))
        return verdict.value, lib().r0h_receipt_verify_reason(verdict.value).decode(), seg.value, sv.value

    def verify_image(self, blob, control_roots, image_blob, image_id, image_control_root=None):
        """`receipt.verify(image_id)` for a trace-circuit session WITHOUT the ELF (r0h_receipt_verify_image): the receipt's image proof
        stands for the program image.  -> (verdict, reason, segment at fault, seal verdict)"""
        b, pb = _u32arr(blob)
        ib, pib = _u32arr(image_blob)
        table = np.zeros(max(1, len(control_roots)) * 9, dtype=np.uint32)
        for k, (size, root) in enumerate(sorted(control_roots.items())):
            table[9 * k] = size
            table[9 * k + 1:9 * k + 9] = root
        proot = None
        if image_control_root is not None:
            root, proot = _u32arr(image_control_root)
        verdict, seg, sv = _c.c_int(-1), _sz(0), _c.c_int(0)
        _check(lib().r0h_receipt_verify_image(self.handle, pb, b.size, table.ctypes.data_as(_vp), len(control_roots), pib, ib.size, proot,
                                              None if image_id is None else bytes(image_id), ctypes.byref(verdict), ctypes.byref(seg), ctypes.byref(sv)))
        return verdict.value, lib().r0h_receipt_verify_reason(verdict.value).decode(), seg.value, sv.value

    @property
    def image_proof(self):
        """the image proof's seal (None when the receipt carries none)"""
        p, n = _vp(), _sz(0)
        _check(lib().r0h_receipt_image_proof(self.handle, ctypes.byref(p), ctypes.byref(n)))
        return np.frombuffer(ctypes.string_at(p, n.value * 4), dtype=np.uint32).copy() if n.value else None

    @image_proof.setter
    def image_proof(self, seal):
        a, pa = _u32arr(seal if seal is not None else np.zeros(0, dtype=np.uint32))
        _check(lib().r0h_receipt_set_image_proof(self.handle, pa, a.size))

    @property
    def kind(self):
        return "Fake" if lib().r0h_receipt_kind(self.handle) == 0 else "Composite"

    @property
    def journal(self):
        p, n = _vp(), _sz(0)
        _check(lib().r0h_receipt_journal(self.handle, ctypes.byref(p), ctypes.byref(n)))
        return ctypes.string_at(p, n.value) if n.value else b""

    def seals(self):
        out = []
        for i in range(lib().r0h_receipt_n_segments(self.handle)):
            p, n, idx = _vp(), _sz(0), _u32(0)
            _check(lib().r0h_receipt_segment(self.handle, i, ctypes.byref(p), ctypes.byref(n), ctypes.byref(idx)))
            out.append((idx.value, np.frombuffer(ctypes.string_at(p, n.value * 4), dtype=np.uint32).copy()))
        return out

    def to_json(self):
        p = _vp()
        _check(lib().r0h_receipt_to_json(self.handle, ctypes.byref(p)))
        text = ctypes.cast(p, ctypes.c_char_p).value.decode("utf-8")
        lib().r0h_free_error(p)
        return text

    def close(self):
        if self.handle:
            lib().r0h_receipt_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Session:
    """A session between its two phases (r0h_session_*): this rank's segments are committed; the proofs wait for the challenge that
    the records of ALL segments determine."""

    def __init__(self, handle):
        self.handle = handle

    @property
    def n_segments(self):
        return lib().r0h_session_n_segments(self.handle)

    def records(self):
        """(indices, records [k, SESSION_RECORD_WORDS]) of this rank's segments"""
        cap = self.n_segments + 1
        idx, rec, n = np.zeros(cap, np.uint32), np.zeros((cap, SESSION_RECORD_WORDS), np.uint32), _sz(0)
        _check(lib().r0h_session_records(self.handle, idx.ctypes.data_as(_vp), rec.ctypes.data_as(_vp), cap, ctypes.byref(n)))
        return idx[:n.value].copy(), rec[:n.value].copy()

    def finish(self, all_records):
        """all_records: [n_segments, SESSION_RECORD_WORDS] in index order.  Returns (Receipt of this rank's segments, image id, cycles)."""
        rec = np.ascontiguousarray(all_records, dtype=np.uint32).reshape(-1, SESSION_RECORD_WORDS)
        h, image_id, cycles = _vp(), (ctypes.c_uint8 * 32)(), _u64(0)
        _check(lib().r0h_session_finish(self.handle, rec.ctypes.data_as(_vp), rec.shape[0], ctypes.byref(h), image_id, ctypes.byref(cycles)))
        return Receipt(h), bytes(image_id), cycles.value

    def close(self):
        if self.handle:
            lib().r0h_session_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Hal:
    """One context = one GPU + one stream (risc0 `Hal` + `CircuitHal` + segment prover, by name)."""

    def __init__(self, device=0):
        self.ctx = _vp()
        _check(lib().r0h_ctx_create(device, ctypes.byref(self.ctx)))  # (lib() has put torch's HIP runtime first: either import order works)

    def close(self):
        if self.ctx:
            _check(lib().r0h_ctx_destroy(self.ctx))
            self.ctx = None

    def sync(self):
        _check(lib().r0h_sync(self.ctx))

    # ---- buffers
    def alloc(self, words):
        h = _vp()
        _check(lib().r0h_buf_alloc(self.ctx, int(words) * 4, ctypes.byref(h)))
        return Buf(self, h, int(words))

    def copy_from(self, array):
        a = np.ascontiguousarray(array, dtype=np.uint32).reshape(-1)
        b = self.alloc(max(a.size, 1))
        if a.size:
            b.upload(a)
        b.words = a.size
        return b

    def wrap(self, device_ptr, words):
        h = _vp()
        _check(lib().r0h_buf_wrap(self.ctx, device_ptr, int(words) * 4, ctypes.byref(h)))
        return Buf(self, h, int(words))

    def zero(self, buf):
        _check(lib().r0h_buf_zero(self.ctx, buf.handle))

    # ---- Hal ops
    def batch_interpolate_ntt(self, io, count, po2):
        _check(lib().r0h_batch_interpolate_ntt(self.ctx, io.handle, count, po2))

    def batch_expand_into_evaluate_ntt(self, out, inp, count, in_po2, expand_bits):
        _check(lib().r0h_batch_expand_into_evaluate_ntt(self.ctx, out.handle, inp.handle, count, in_po2, expand_bits))

    def batch_bit_reverse(self, io, count, po2):
        _check(lib().r0h_batch_bit_reverse(self.ctx, io.handle, count, po2))

    def zk_shift(self, io, count, po2):
        _check(lib().r0h_zk_shift(self.ctx, io.handle, count, po2))

    def batch_interpolate_ntt_zk_shift(self, io, count, po2):
        """batch_interpolate_ntt + zk_shift in one call (the shift fused into the transform's last pass)."""
        _check(lib().r0h_batch_interpolate_ntt_zk_shift(self.ctx, io.handle, count, po2))

    def poseidon2_set_consts(self, rc, diag_m1):
        a, pa = _u32arr(rc)
        b, pb = _u32arr(diag_m1)
        assert a.size == 24 * 29 and b.size == 24
        _check(lib().r0h_poseidon2_set_consts(self.ctx, pa, pb))

    def hash_rows(self, digests, matrix, rows, cols):
        _check(lib().r0h_hash_rows(self.ctx, digests.handle, matrix.handle, rows, cols))

    def hash_fold(self, nodes, output_size):
        _check(lib().r0h_hash_fold(self.ctx, nodes.handle, output_size))

    def merkle_build(self, nodes, matrix, rows, cols):
        _check(lib().r0h_merkle_build(self.ctx, nodes.handle, matrix.handle, rows, cols))

    def batch_evaluate_any(self, coeffs, po2, which, xs, out):
        w, pw = _u32arr(which)
        x, px = _u32arr(xs)
        assert x.size == 4 * w.size
        _check(lib().r0h_batch_evaluate_any(self.ctx, coeffs.handle, po2, pw, px, w.size, out.handle))

    def mix_poly_coeffs(self, combos, mix_start, mix, inp, combo_of, po2):
        ms, pms = _u32arr(mix_start)
        m, pm = _u32arr(mix)
        co, pco = _u32arr(combo_of)
        _check(lib().r0h_mix_poly_coeffs(self.ctx, combos.handle, pms, pm, inp.handle, pco, co.size, po2))

    def eltwise_add_elem(self, out, a, b, n):
        _check(lib().r0h_eltwise_add_elem(self.ctx, out.handle, a.handle, b.handle, n))

    def eltwise_copy_elem(self, out, inp, n):
        _check(lib().r0h_eltwise_copy_elem(self.ctx, out.handle, inp.handle, n))

    def eltwise_sum_extelem(self, out, inp, count, n):
        _check(lib().r0h_eltwise_sum_extelem(self.ctx, out.handle, inp.handle, count, n))

    def eltwise_zeroize_elem(self, io, n):
        _check(lib().r0h_eltwise_zeroize_elem(self.ctx, io.handle, n))

    def gather_sample(self, dst, src, idx, size, stride):
        _check(lib().r0h_gather_sample(self.ctx, dst.handle, src.handle, idx, size, stride))

    def scatter(self, into, index, offsets, values, n_index):
        _check(lib().r0h_scatter(self.ctx, into.handle, index.handle, offsets.handle, values.handle, n_index))

    # ---- the Hal trait's operand placement: device buffers for which / xs / combos, host slices for scatter
    def batch_evaluate_any_buf(self, coeffs, po2, which_buf, xs_buf, n_eval, out):
        _check(lib().r0h_batch_evaluate_any_buf(self.ctx, coeffs.handle, po2, which_buf.handle, xs_buf.handle, n_eval, out.handle))

    def mix_poly_coeffs_buf(self, combos, mix_start, mix, inp, combo_buf, input_count, po2):
        ms, pms = _u32arr(mix_start)
        m, pm = _u32arr(mix)
        _check(lib().r0h_mix_poly_coeffs_buf(self.ctx, combos.handle, pms, pm, inp.handle, combo_buf.handle, input_count, po2))

    def scatter_slices(self, into, index, offsets, values):
        i, pi = _u32arr(index)
        o, po = _u32arr(offsets)
        v, pv = _u32arr(values)
        assert o.size == v.size
        _check(lib().r0h_scatter_slices(self.ctx, into.handle, pi, i.size, po, pv, o.size))

    def hash_fold_io(self, io, input_size, output_size):
        _check(lib().r0h_hash_fold_io(self.ctx, io.handle, input_size, output_size))

    def fri_fold(self, out, inp, mix, n_out):
        m, pm = _u32arr(mix)
        _check(lib().r0h_fri_fold(self.ctx, out.handle, inp.handle, pm, n_out))

    def prefix_products(self, io, n):
        _check(lib().r0h_prefix_products(self.ctx, io.handle, n))

    def poly_divide(self, poly, n, z):
        zz, pz = _u32arr(z)
        rem = np.zeros(4, dtype=np.uint32)
        _check(lib().r0h_poly_divide(self.ctx, poly.handle, n, pz, rem.ctypes.data_as(_vp)))
        return rem

    # ---- CircuitHal + sequencer
    def load_circuit(self, blob, code_object_path=None):
        a, p = _u32arr(blob)
        h = _vp()
        path = code_object_path.encode() if code_object_path else None
        _check(lib().r0h_circuit_load(self.ctx, p, a.size, path, ctypes.byref(h)))
        return Circuit(self, h, a)

    def witgen(self, circuit, po2, seed, globals_in=None):
        """Synthetic witness; with globals_in the caller's public inputs are planted (r0h_witgen_public)."""
        n = 1 << po2
        code = self.alloc(circuit.group_size[GROUP_CODE] * n)
        data = self.alloc(circuit.group_size[GROUP_DATA] * n)
        if globals_in is not None:
            g, pg = _u32arr(globals_in)
            if g.size != circuit.n_global:
                raise R0HipError("witgen: %d public inputs given, the circuit has %d" % (g.size, circuit.n_global))
            _check(lib().r0h_witgen_public(self.ctx, circuit.handle, po2, seed, pg, code.handle, data.handle))
            return code, data, g.copy()
        glob = np.zeros(max(circuit.n_global, 1), dtype=np.uint32)
        _check(lib().r0h_witgen(self.ctx, circuit.handle, po2, seed, code.handle, data.handle, glob.ctypes.data_as(_vp)))
        return code, data, glob[:circuit.n_global]

    def trace_witgen(self, rows, bounds, po2, claim_globals=None, into=None, number=1, closing=True, idle_pc=0, circuit=None):
        """DATA group of the trace circuit expanded ON THE DEVICE from the compact preflight rows (r0h_trace_witgen): rows [n, 18]
        and bounds [m, 8] uint32 (Vm.preflight_arrays); number / closing: the segment's number in its session and whether its
        boundary rows close it.  With `circuit` (the loaded trace circuit) the lookup tables' multiplicity columns are filled too
        (r0h_logup_multiplicities); without, they stay zero.  Returns (Buf of TRACE_COLUMNS * 2^po2 words, the TRACE_GLOBALS public
        inputs -- the late ones zero)."""
        rows = np.ascontiguousarray(rows, dtype=np.uint32).reshape(-1, 18)
        bounds = np.ascontiguousarray(bounds, dtype=np.uint32).reshape(-1, 8)
        data = into if into is not None else self.alloc(TRACE_COLUMNS << po2)
        glob = np.zeros(TRACE_GLOBALS, dtype=np.uint32)
        seg = TraceSegment(number, 1 if closing else 0, idle_pc, 0)
        try:
            _check(lib().r0h_trace_witgen(self.ctx, rows.ctypes.data_as(_vp), rows.shape[0], bounds.ctypes.data_as(_vp), bounds.shape[0], po2, ctypes.byref(seg),
                                          data.handle, glob.ctypes.data_as(_vp)))
            if circuit is not None:
                _check(lib().r0h_logup_multiplicities(self.ctx, circuit.handle, po2, data.handle, glob.ctypes.data_as(_vp)))
        except R0HipError:
            if into is None:
                data.free()
            raise
        if claim_globals is not None:
            glob[:8] = claim_globals
        return data, glob

    def witgen_into(self, circuit, po2, seed, code, data):
        """Regenerate the synthetic witness of another segment into existing buffers; returns its public inputs."""
        glob = np.zeros(max(circuit.n_global, 1), dtype=np.uint32)
        _check(lib().r0h_witgen(self.ctx, circuit.handle, po2, seed, code.handle, data.handle, glob.ctypes.data_as(_vp)))
        return glob[:circuit.n_global]

    def accum(self, circuit, po2, code, data, mix):
        n = 1 << po2
        m, pm = _u32arr(mix)
        out = self.alloc(circuit.group_size[GROUP_ACCUM] * n)
        _check(lib().r0h_accum(self.ctx, circuit.handle, po2, code.handle, data.handle, pm, out.handle))
        return out

    def accum_public(self, circuit, po2, code, data, glob, mix):
        """r0h_accum for a circuit whose accumulation reads public inputs (the log-derivative argument of the trace circuit)"""
        n = 1 << po2
        g, pg = _u32arr(glob)
        m, pm = _u32arr(mix)
        out = self.alloc(circuit.group_size[GROUP_ACCUM] * n)
        _check(lib().r0h_accum_public(self.ctx, circuit.handle, po2, code.handle, data.handle, pg, pm, out.handle))
        return out

    def logup_totals(self, circuit, po2, code, data, glob):
        """the public inputs with the totals of the accumulators that run under public challenges filled in (r0h_logup_totals)"""
        g = np.ascontiguousarray(glob, dtype=np.uint32).copy()
        _check(lib().r0h_logup_totals(self.ctx, circuit.handle, po2, code.handle, data.handle, g.ctypes.data_as(_vp)))
        return g

    def prefix_sums(self, io, n):
        _check(lib().r0h_prefix_sums(self.ctx, io.handle, n))

    def eval_check(self, circuit, po2, eval_accum, eval_code, eval_data, glob, mix, poly_mix):
        g, pg = _u32arr(glob)
        m, pm = _u32arr(mix)
        q, pq = _u32arr(poly_mix)
        check = self.alloc(16 << po2)
        _check(lib().r0h_eval_check(self.ctx, circuit.handle, po2, eval_accum.handle, eval_code.handle, eval_data.handle,
                                    pg, pm, pq, check.handle))
        return check

    def prove_segment(self, circuit, po2, code, data, glob, seal_capacity_words=1 << 20):
        """code: the CODE witness columns (Buf) or their commitment (CodeCommit, r0h_prove_segment_committed) -- same seal."""
        g, pg = _u32arr(glob)
        seal = np.empty(seal_capacity_words, dtype=np.uint32)
        n = _sz(0)
        fn = lib().r0h_prove_segment_committed if isinstance(code, CodeCommit) else lib().r0h_prove_segment
        _check(fn(self.ctx, circuit.handle, po2, code.handle, data.handle, pg, seal.ctypes.data_as(_vp), seal.size, ctypes.byref(n)))
        return seal[:n.value].copy()

    def code_commit(self, circuit, po2, code=None):
        """Commit the CODE group of `circuit` at 2^po2 rows once (r0h_code_commit_new).  Synthetic circuits regenerate their fixed
        CODE columns; pass `code` for a circuit that brings its own."""
        own = None
        if code is None:
            code, own, _ = self.witgen(circuit, po2, seed=0)
        h = _vp()
        try:
            _check(lib().r0h_code_commit_new(self.ctx, code.handle, circuit.group_size[GROUP_CODE], po2, ctypes.byref(h)))
        finally:
            if own is not None:
                own.free()
                code.free()
        return CodeCommit(h, po2)

    def code_root(self, circuit, po2, code=None):
        """Control root of the circuit at 2^po2 rows: Merkle root of its committed CODE group (r0h_code_root).  Synthetic
        circuits regenerate their fixed CODE columns; pass `code` for a circuit that brings its own."""
        own = None
        if code is None:
            code, own, _ = self.witgen(circuit, po2, seed=0)
        out = np.zeros(8, dtype=np.uint32)
        try:
            _check(lib().r0h_code_root(self.ctx, code.handle, circuit.group_size[GROUP_CODE], po2, out.ctypes.data_as(_vp)))
        finally:
            if own is not None:
                own.free()
                code.free()
        return out

    def prove_elf(self, circuit, elf, input_words, segment_po2=20, max_cycles=0, part=0, parts=1):
        """`default_prover().prove(env, elf)` (r0h_prove_elf): returns (Receipt, image id, guest cycles).  max_cycles = 0 selects the
        library's session limit (2^32 cycles); with circuits/trace.r0c every seal attests the segment it stands for.  parts > 1
        (r0h_prove_elf_part): this rank's share of a session proved on several GPUs -- segments part, part + parts, ...; the receipt
        holds those only (Receipt.merge puts the ranks' receipts together)."""
        elf = bytes(elf)
        w, pw = _u32arr(input_words if len(input_words) else [0])
        h, image_id, cycles = _vp(), (ctypes.c_uint8 * 32)(), _u64(0)
        _check(lib().r0h_prove_elf_part(self.ctx, circuit.handle, elf, len(elf), pw, len(input_words), segment_po2, max_cycles, part, parts, ctypes.byref(h),
                                        image_id, ctypes.byref(cycles)))
        return Receipt(h), bytes(image_id), cycles.value

    def session_begin(self, circuit, elf, input_words, segment_po2=20, max_cycles=0, part=0, parts=1):
        """Phase 1 of a session some ranks share (r0h_session_begin): executes the guest and commits this rank's segments.  Returns a
        Session: .records() is what its segments contribute to the session challenge, .finish(all records) the rank's receipt."""
        elf = bytes(elf)
        w, pw = _u32arr(input_words if len(input_words) else [0])
        h = _vp()
        _check(lib().r0h_session_begin(self.ctx, circuit.handle, elf, len(elf), pw, len(input_words), segment_po2, max_cycles, part, parts, ctypes.byref(h)))
        return Session(h)

    def set_session_resident_limit(self, n_bytes):
        """bytes of committed DATA evaluations a session on this context keeps between its phases (r0h_ctx_set_session_resident_limit;
        0 = an eighth of the device's memory); segments beyond it are evaluated again when their proofs are finished -- same seals"""
        _check(lib().r0h_ctx_set_session_resident_limit(self.ctx, n_bytes))

    def set_image_circuit(self, image_circuit):
        """sessions on this context attach an image proof to their receipts from now on (r0h_ctx_set_image_circuit; None: stop)"""
        _check(lib().r0h_ctx_set_image_circuit(self.ctx, image_circuit.handle if image_circuit is not None else None))
        self._image_circuit = image_circuit  # (kept alive)

    def prove_image(self, image_circuit, elf, challenge):
        """One image proof under a session's challenge (r0h_prove_image): the seal's public inputs are the image's digest, the
        challenge and the image's side of the session's balance."""
        buf = (ctypes.c_uint8 * len(elf)).from_buffer_copy(elf)
        ch, pch = _u32arr(challenge)
        assert ch.size == 16
        seal = np.empty(1 << 18, dtype=np.uint32)
        n = _sz(0)
        _check(lib().r0h_prove_image(self.ctx, image_circuit.handle, buf, len(elf), pch, seal.ctypes.data_as(_vp), seal.size, ctypes.byref(n)))
        return seal[:n.value].copy()

    def last_session_stats(self):
        """Stage timing of the last prove_elf on this context (r0h_last_session_stats)."""
        st = SessionStats()
        _check(lib().r0h_last_session_stats(self.ctx, ctypes.byref(st)))
        return {name: getattr(st, name) for name, _ in SessionStats._fields_}

    def proof_begin(self, circuit, po2, code, data, glob):
        """Commit CODE and DATA; returns (proof handle, accumulation mix words)."""
        g, pg = _u32arr(glob)
        mix = np.zeros(max(circuit.n_mix, 1), dtype=np.uint32)
        h = _vp()
        fn = lib().r0h_proof_begin_committed if isinstance(code, CodeCommit) else lib().r0h_proof_begin
        _check(fn(self.ctx, circuit.handle, po2, code.handle, data.handle, pg, mix.ctypes.data_as(_vp), ctypes.byref(h)))
        return h, mix[:circuit.n_mix]

    def proof_finish(self, proof, accum, seal_capacity_words=1 << 20):
        seal = np.empty(seal_capacity_words, dtype=np.uint32)
        n = _sz(0)
        _check(lib().r0h_proof_finish(proof, accum.handle, seal.ctypes.data_as(_vp), seal.size, ctypes.byref(n)))
        return seal[:n.value].copy()

    def proof_abort(self, proof):
        _check(lib().r0h_proof_abort(proof))

    def kernel_timing(self, enable=True):
        _check(lib().r0h_kernel_timing(self.ctx, 1 if enable else 0))

    def kernel_stats(self):
        import json
        buf = ctypes.create_string_buffer(1 << 16)
        _check(lib().r0h_kernel_stats(self.ctx, buf, len(buf)))
        return json.loads(buf.value.decode())

    def last_profile(self):
        names = _c.POINTER(_cp)()
        ms = _c.POINTER(_c.c_float)()
        n = _u32(0)
        _check(lib().r0h_last_profile(self.ctx, ctypes.byref(names), ctypes.byref(ms), ctypes.byref(n)))
        return [(names[i].decode(), float(ms[i])) for i in range(n.value)]
