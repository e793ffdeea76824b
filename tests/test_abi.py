"""CPU checks of the drop-in boundary: libr0hip.so loads and exports exactly what include/r0hip.h declares
(no compute is attempted without a GPU), the generated eval_check source is well-formed, and the product path
refuses to run without a device instead of falling back to anything."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, circuit_path


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "r0hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(r0h_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import hyperfridge_r0_amd as r0
    lib = r0.lib()
    names = declared_symbols()
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), "libr0hip.so does not export " + n
    assert names == r0.EXPORTED_SYMBOLS, "the ctypes harness and include/r0hip.h disagree"
    assert b"gfx950" in lib.r0h_version()


def test_product_does_not_link_or_import_the_oracle():
    so = open(os.path.join(ROOT, "hyperfridge-r0_amd", "libr0hip.so"), "rb").read()
    assert b"liborc" not in so and b"orc_prove_segment" not in so
    for dirpath, _, files in os.walk(os.path.join(ROOT, "hyperfridge-r0_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")) or f == "Makefile":
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle/" not in text and "liborc" not in text and "orc_" not in text, f


def test_no_device_is_an_error_not_a_fallback():
    import hyperfridge_r0_amd as r0
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        pytest.skip("a GPU is present")
    with pytest.raises(r0.R0HipError):
        r0.Hal(0)


def test_eval_check_codegen_is_deterministic_and_complete():
    import hyperfridge_r0_amd as r0
    blob = np.fromfile(circuit_path("small"), dtype=np.uint32)
    src = r0.emit_eval_check_source(blob)
    assert src == r0.emit_eval_check_source(blob)
    kernels = re.findall(r"void (eval_check_\d+)\(", src)
    assert kernels and kernels == ["eval_check_%d" % i for i in range(len(kernels))]
    # every flattened constraint term appears exactly once
    n_terms = int(re.search(r"// terms: (\d+)", src).group(1))
    assert len(re.findall(r"const u32 w\d+ = ", src)) == n_terms
    # a corrupted blob is rejected with a message, not a crash
    bad = blob.copy()
    bad[0] ^= 1
    with pytest.raises(r0.R0HipError):
        r0.emit_eval_check_source(bad)
    bad = blob[:-7]
    with pytest.raises(r0.R0HipError):
        r0.emit_eval_check_source(bad)


def test_blob_parser_survives_random_corruption():
    """Host-only fuzz of the circuit loader's validation (r0h_circuit_emit_hip parses and plans without a GPU): a mutated
    blob must either be rejected with a message or produce source -- never crash, never hang."""
    import hyperfridge_r0_amd as r0
    blob = np.fromfile(circuit_path("tiny"), dtype=np.uint32)
    rng = np.random.default_rng(2024)
    rejected = accepted = 0
    for trial in range(300):
        bad = blob.copy()
        for _ in range(int(rng.integers(1, 4))):
            pos = int(rng.integers(0, bad.size))
            mode = int(rng.integers(0, 3))
            bad[pos] = [int(rng.integers(0, 2**32)), int(bad[pos]) ^ (1 << int(rng.integers(0, 32))), int(rng.integers(0, 64))][mode]
        if trial % 10 == 0:
            bad = bad[:int(rng.integers(3, bad.size))]
        try:
            r0.emit_eval_check_source(bad)
            accepted += 1
        except r0.R0HipError as e:
            assert str(e)
            rejected += 1
    assert rejected > 50 and accepted + rejected == 300


def test_header_is_plain_c_and_a_c_program_can_drive_the_library(tmp_path):
    """include/r0hip.h must compile as C99 (no C++ in the boundary) and a C caller must be able to link and use the host-only
    entry points: version, serde framing, the receipt parser and the seal verifier on the frozen golden seal."""
    import subprocess
    seal = np.load(os.path.join(ROOT, "tests", "golden", "seal_tiny_po2_9_seed_1.npy"))
    seal_path, blob_path = str(tmp_path / "seal.bin"), circuit_path("tiny")
    seal.tofile(seal_path)
    src = tmp_path / "caller.c"
    src.write_text(r'''
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "r0hip.h"
#include "r0hip_circuit.h"
static uint32_t* slurp(const char* path, size_t* words) {
  FILE* f = fopen(path, "rb"); if (!f) return NULL;
  fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
  uint32_t* p = (uint32_t*)malloc((size_t)n); *words = (size_t)n / 4;
  if (fread(p, 1, (size_t)n, f) != (size_t)n) return NULL;
  fclose(f); return p;
}
int main(int argc, char** argv) {
  size_t nb, ns, len = 0, off = 0, slen = 0; int verdict = -1; uint32_t po2 = 0; uint8_t frame[16];
  uint32_t* blob = slurp(argv[1], &nb); uint32_t* seal = slurp(argv[2], &ns);
  if (!blob || !seal || blob[0] != R0H_BLOB_MAGIC) return 2;
  const char* err = r0h_verify_seal(blob, nb, NULL, NULL, seal, ns, &verdict, &po2);
  if (err) { fprintf(stderr, "%s\n", err); r0h_free_error(err); return 3; }
  if (r0h_serde_encode_str((const uint8_t*)"abcde", 5, frame, sizeof frame, &len) || len != 12) return 4;
  if (r0h_serde_decode_str(frame, len, &off, &slen, NULL) || slen != 5 || memcmp(frame + off, "abcde", 5)) return 5;
  r0h_receipt* rc = NULL;
  const char* json = "{\"inner\":\"Fake\",\"journal\":{\"bytes\":[1,0,0,0,65,0,0,0]}}";
  if (r0h_receipt_parse(json, strlen(json), &rc) || r0h_receipt_kind(rc) != R0H_RECEIPT_FAKE) return 6;
  r0h_receipt_free(rc);
  printf("%s verdict=%d (%s) po2=%u\n", r0h_version(), verdict, r0h_verify_reason(verdict), po2);
  return verdict == R0H_VERIFY_OK ? 0 : 1;
}
''')
    exe = str(tmp_path / "caller")
    libdir = os.path.join(ROOT, "hyperfridge-r0_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src), "-o", exe,
                           "-L", libdir, "-lr0hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe, blob_path, seal_path], capture_output=True, text=True)
    assert out.returncode == 0 and "verdict=0 (ok) po2=9" in out.stdout, out.stdout + out.stderr


def test_rust_bindings_are_generated_from_the_header_and_name_the_same_symbols():
    """bindings/r0hip_sys.rs (the `extern "C"` block a Rust `HipHal` links against) is generated from include/r0hip.h by
    tools/gen_rust_bindings.py; it must be fresh, declare exactly the header's functions -- which the library exports -- and every
    parameter must have come through with a Rust type (no rustc exists here to compile it, so the shape is checked textually)."""
    import subprocess
    import sys
    assert subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_bindings.py"), "--check"]).returncode == 0, "stale bindings"
    rs = open(os.path.join(ROOT, "bindings", "r0hip_sys.rs")).read()
    fns = re.findall(r"pub fn (r0h_\w+)\(([^)]*)\)( -> [^;]+)?;", rs)
    assert sorted(f[0] for f in fns) == declared_symbols()
    import hyperfridge_r0_amd as r0
    lib = r0.lib()
    for name, args, ret in fns:
        assert hasattr(lib, name)
        for a in filter(None, (x.strip() for x in args.split(","))):
            nm, ty = a.split(": ")
            assert re.fullmatch(r"(\*(const|mut) )*(u8|u32|u64|usize|f32|c_int|c_char|c_void|r0h_\w+)", ty), (name, a)
        if ret:
            assert re.fullmatch(r" -> (\*(const|mut) )*(u32|u64|usize|c_int|c_char|c_void)", ret), (name, ret)
    # spot checks of the translation rules
    assert "pub fn r0h_ctx_create(device: c_int, out: *mut *mut r0h_ctx) -> *const c_char;" in rs
    assert "pub fn r0h_buf_device_ptr(buf: *const r0h_buf) -> *mut c_void;" in rs
    assert "pub fn r0h_free_error(msg: *const c_char);" in rs
    assert "pub fn r0h_claim_digest(claim: *const r0h_receipt_claim, digest_out: *mut u8) -> *const c_char;" in rs
    # every type a prototype names is declared in the file: the opaque handles and the plain structs, field for field
    declared = set(re.findall(r"pub struct (r0h_\w+)", rs))
    assert set(re.findall(r"\b(r0h_\w+)\b", " ".join(a + (r or "") for _, a, r in fns))) - {n for n, _, _ in fns} <= declared
    assert "pub struct r0h_system_state { pub pc: u32, pub merkle_root: [u8; 32] }" in rs
    assert "pub struct r0h_vm_limits { pub segment_po2: u32, pub page_in_cycles: u32, pub page_out_cycles: u32, pub keep_trace: u32, pub max_cycles: u64, pub boundary_rows: u32, pub reserved: u32 }" in rs
    assert "pub struct r0h_ctx { _private: [u8; 0] }" in rs and "pub struct r0h_vm { _private: [u8; 0] }" in rs
    # INTEGRATION.md points at the generated file instead of carrying its own (partial) copy
    assert "bindings/r0hip_sys.rs" in open(os.path.join(ROOT, "INTEGRATION.md")).read()
