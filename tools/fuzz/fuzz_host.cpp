// libFuzzer + AddressSanitizer harness for the host-only parsers of libr0hip.so: everything here reads bytes that come from
// outside (a bank's EBICS response, PEM keys, a receipt file, a seal, a circuit blob, an ELF) and runs without a GPU.
// Build and run: tools/fuzz/run.sh [seconds]   (host-only compile of the .hip sources with ROCm's clang; no device code involved)
// The first input byte selects the target; tools/fuzz/make_seeds.py writes one seed per target from tests/golden/.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/r0hip.h"

static std::vector<uint8_t> slurp(const std::string& path) {
  std::vector<uint8_t> v;
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) { fprintf(stderr, "fuzz_host: cannot read %s (set R0H_FUZZ_ROOT to the repository root)\n", path.c_str()); abort(); }
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  v.resize(n > 0 ? (size_t)n : 0);
  if (n > 0 && fread(v.data(), 1, v.size(), f) != v.size()) abort();
  fclose(f);
  return v;
}
static void drop(const char* err) { if (err) r0h_free_error(err); }
static bool good(const char* err) { drop(err); return err == nullptr; }

struct Fixtures {
  std::vector<uint8_t> xml, pub_bank, pub_client, pub_witness, client_pem, witness_pem, tx_raw, witness_hex, circuit, seal_npy;
  std::vector<uint32_t> blob, seal, image_blob;
  Fixtures() {
    const char* env = getenv("R0H_FUZZ_ROOT");
    const std::string root = env ? env : ".";
    const std::string g = root + "/tests/golden/camt53/";
    xml = slurp(g + "response.xml"); pub_bank = slurp(g + "pub_bank.pem"); pub_client = slurp(g + "pub_client.pem");
    pub_witness = slurp(g + "pub_witness.pem"); client_pem = slurp(g + "client.pem"); witness_pem = slurp(g + "witness.pem");
    tx_raw = slurp(g + "test.xml-TransactionKeyDecrypt.bin"); witness_hex = slurp(g + "test.xml-Witness.hex");
    circuit = slurp(root + "/circuits/tiny.r0c");
    blob.assign((const uint32_t*)circuit.data(), (const uint32_t*)circuit.data() + circuit.size() / 4);
    {
      const std::vector<uint8_t> ic = slurp(root + "/circuits/image.r0c");
      image_blob.assign((const uint32_t*)ic.data(), (const uint32_t*)ic.data() + ic.size() / 4);
    }
    seal_npy = slurp(root + "/tests/golden/seal_tiny_po2_9_seed_1.npy");  // .npy v1: magic, version, u16 header length, header, then the words
    if (seal_npy.size() < 10) abort();
    const size_t off = 10 + (seal_npy[8] | (size_t)seal_npy[9] << 8);
    if (seal_npy.size() <= off) abort();
    seal.resize((seal_npy.size() - off) / 4);
    memcpy(seal.data(), seal_npy.data() + off, seal.size() * 4);
  }
};
static const Fixtures& fx() { static Fixtures f; return f; }

static void walk_ebics(r0h_ebics* e) {
  for (int which = 0; which <= 8; which++) {
    const uint8_t* p; size_t n;
    drop(r0h_ebics_part(e, which, &p, &n));
  }
  int ok = 0;
  drop(r0h_ebics_check_digest(e, &ok));
  drop(r0h_ebics_verify_bank_signature(e, (const char*)fx().pub_bank.data(), fx().pub_bank.size(), &ok));
  uint8_t key[16] = {0};
  drop(r0h_ebics_check_transaction_key(e, (const char*)fx().pub_client.data(), fx().pub_client.size(), fx().tx_raw.data(), fx().tx_raw.size(), key, &ok));
  if (good(r0h_ebics_decrypt_order_data(e, key))) {
    for (size_t i = 0; i < r0h_ebics_n_documents(e); i++) {
      const char* name; const uint8_t* data; size_t n;
      drop(r0h_ebics_document(e, i, &name, &data, &n));
    }
  }
  drop(r0h_ebics_verify_witness(e, (const char*)fx().pub_witness.data(), fx().pub_witness.size(), (const char*)fx().witness_hex.data(), fx().witness_hex.size(), &ok));
  r0h_env* env = nullptr;
  if (good(r0h_ebics_env_inputs(e, (const char*)fx().pub_bank.data(), fx().pub_bank.size(), (const char*)fx().client_pem.data(), fx().client_pem.size(), fx().tx_raw.data(),
                              fx().tx_raw.size(), "CH4308307000289537312", "host", (const char*)fx().witness_hex.data(), fx().witness_hex.size(),
                              (const char*)fx().pub_witness.data(), fx().pub_witness.size(), "", &env)))
    r0h_env_free(env);
}

static void run_vm(r0h_vm* vm) {
  uint32_t input[8] = {1, 2, 3, 4, 5, 6, 7, 8};
  drop(r0h_vm_set_input(vm, input, 8));
  r0h_vm_limits lim = {10, 16, 16, 1, 20000, 1, 0};  // trace kept, boundary rows charged: the r0h_prove_elf configuration
  int kind = 0; uint32_t code = 0;
  drop(r0h_vm_run(vm, &lim, &kind, &code));
  for (size_t i = 0; i < r0h_vm_n_segments(vm); i++) {
    r0h_vm_segment seg; r0h_receipt_claim cl; const r0h_preflight_row* rows; size_t n; const r0h_preflight_bound* bounds; size_t nb;
    drop(r0h_vm_segment_info(vm, i, &seg));
    drop(r0h_vm_preflight(vm, i, &rows, &n));
    drop(r0h_vm_boundary(vm, i, &bounds, &nb));
    drop(r0h_vm_segment_claim(vm, i, &cl));
    if (i == 0 && n > 0) {  // the trace circuit's witness from these rows
      uint32_t po2 = 4;
      while (((size_t)1 << po2) < n + nb) po2++;
      std::vector<uint32_t> w((size_t)R0H_TRACE_COLUMNS << po2);
      uint32_t g[R0H_TRACE_GLOBALS];
      drop(r0h_vm_trace_witness(vm, i, po2, w.data(), g));
      drop(r0h_vm_trace_witness(vm, i, po2 - 1, w.data(), g));  // too small: an error, not an overrun
    }
  }
  const uint8_t* j; size_t nj;
  drop(r0h_vm_journal(vm, &j, &nj));
}

extern "C" int LLVMFuzzerTestOneInput(const uint8_t* data, size_t size) {
  if (size < 1) return 0;
  const int target = data[0] % 9;
  const uint8_t* d = data + 1;
  const size_t n = size - 1;
  switch (target) {
    case 0: {  // an EBICS response
      r0h_ebics* e = nullptr;
      if (good(r0h_ebics_parse((const char*)d, n, &e))) { walk_ebics(e); r0h_ebics_free(e); }
      break;
    }
    case 1: {  // an RFC 1950 stream
      uint8_t* out = nullptr; size_t on = 0;
      if (good(r0h_zlib_inflate(d, n, &out, &on))) r0h_free_error((const char*)out);
      break;
    }
    case 2: {  // PEM / DER keys and hex signatures against the genuine response
      char *m = nullptr, *ex = nullptr;
      if (good(r0h_rsa_public_key_decimal((const char*)d, n, &m, &ex))) { r0h_free_error(m); r0h_free_error(ex); }
      r0h_ebics* e = nullptr;
      if (!good(r0h_ebics_parse((const char*)fx().xml.data(), fx().xml.size(), &e))) abort();  // the fixture must parse
      int ok = 0;
      drop(r0h_ebics_verify_bank_signature(e, (const char*)d, n, &ok));
      drop(r0h_ebics_verify_witness(e, (const char*)fx().pub_witness.data(), fx().pub_witness.size(), (const char*)d, n, &ok));
      uint8_t raw[512], key[16]; size_t rn = 0;
      drop(r0h_ebics_decrypt_transaction_key(e, (const char*)d, n, raw, sizeof raw, &rn, key, &ok));
      drop(r0h_ebics_check_transaction_key(e, (const char*)fx().pub_client.data(), fx().pub_client.size(), d, n, key, &ok));
      {  // the guest's input words with the fuzzed bytes in each role in turn: a key, the decrypted block, the witness signature
        uint32_t* words = nullptr; size_t nw = 0;
        const char *wk = (const char*)fx().pub_witness.data(), *ck = (const char*)fx().pub_client.data();
        const size_t wn = fx().pub_witness.size(), cn = fx().pub_client.size();
        if (good(r0h_camt53_guest_input(e, (const char*)d, n, ck, cn, wk, wn, raw, 256, (const char*)d, n, "CH00", "h", 1, &words, &nw))) r0h_free_error((const char*)words);
        if (good(r0h_camt53_guest_input(e, ck, cn, ck, cn, wk, wn, d, n, (const char*)d, n, "CH00", "h", 0, &words, &nw))) r0h_free_error((const char*)words);
      }
      r0h_ebics_free(e);
      break;
    }
    case 3: {  // an ELF
      r0h_vm* vm = nullptr;
      if (!good(r0h_vm_new(&vm))) break;
      if (good(r0h_vm_load_elf(vm, d, n))) run_vm(vm);
      { uint8_t id[32]; drop(r0h_compute_image_id(d, n, id)); }
      {  // the image circuit's witness of whatever this file loads as an image
        uint32_t po2 = 0;
        if (good(r0h_image_po2(d, n, &po2)) && po2 <= 14) {
          std::vector<uint32_t> cols((size_t)R0H_IMAGE_COLUMNS << po2), glob(R0H_IMAGE_GLOBALS);
          drop(r0h_image_witness(d, n, po2, cols.data(), glob.data()));
        }
      }
      r0h_vm_free(vm);
      break;
    }
    case 4: {  // raw instruction words
      r0h_vm* vm = nullptr;
      if (!good(r0h_vm_new(&vm))) break;
      std::vector<uint32_t> w(n / 4);
      if (!w.empty()) memcpy(w.data(), d, w.size() * 4);
      if (good(r0h_vm_load(vm, 0x1000, w.data(), w.size())) && good(r0h_vm_set_pc(vm, 0x1000))) run_vm(vm);
      r0h_vm_free(vm);
      break;
    }
    case 5: {  // a receipt file
      r0h_receipt* rc = nullptr;
      if (good(r0h_receipt_parse((const char*)d, n, &rc))) {
        char* js = nullptr;
        if (good(r0h_receipt_to_json(rc, &js))) r0h_free_error(js);
        for (size_t i = 0; i < r0h_receipt_n_segments(rc); i++) {
          const uint32_t* seal; size_t words; uint32_t index; r0h_receipt_claim cl; int has = 0;
          drop(r0h_receipt_segment(rc, i, &seal, &words, &index));
          drop(r0h_receipt_segment_claim(rc, i, &cl, &has));
        }
        int verdict = 0, sv = 0; size_t seg = 0;
        uint32_t roots[9] = {9, 1, 2, 3, 4, 5, 6, 7, 8};
        uint8_t image[32] = {0};
        drop(r0h_receipt_verify(rc, fx().blob.data(), fx().blob.size(), roots, 1, image, &verdict, &seg, &sv));
        if (!fx().image_blob.empty())  // ... and with whatever it carries as an image proof
          drop(r0h_receipt_verify_image(rc, fx().blob.data(), fx().blob.size(), roots, 1, fx().image_blob.data(), fx().image_blob.size(), nullptr, image, &verdict, &seg, &sv));
        { const uint32_t* ip; size_t ipn; drop(r0h_receipt_image_proof(rc, &ip, &ipn)); }
        {  // a receipt merged with itself (segments twice: refused for a composite one) and on its own (complete or not)
          const r0h_receipt* twice[2] = {rc, rc};
          r0h_receipt* merged = nullptr;
          if (good(r0h_receipt_merge(twice, 2, &merged))) r0h_receipt_free(merged);
          merged = nullptr;
          if (good(r0h_receipt_merge(twice, 1, &merged))) {
            char* js2 = nullptr;
            if (good(r0h_receipt_to_json(merged, &js2))) r0h_free_error(js2);
            r0h_receipt_free(merged);
          }
        }
        r0h_receipt_free(rc);
      }
      break;
    }
    case 6: {  // a seal against a genuine circuit
      std::vector<uint32_t> w(n / 4);
      if (!w.empty()) memcpy(w.data(), d, w.size() * 4);
      int verdict = 0; uint32_t po2 = 0, root[8];
      drop(r0h_verify_seal_bound(fx().blob.data(), fx().blob.size(), nullptr, nullptr, w.data(), w.size(), nullptr, &verdict, &po2, root));
      uint32_t dg[8];
      drop(r0h_seal_digest(w.data(), w.size(), dg));
      break;
    }
    case 7: {  // a circuit blob against a genuine seal
      std::vector<uint32_t> w(n / 4);
      if (!w.empty()) memcpy(w.data(), d, w.size() * 4);
      int verdict = 0; uint32_t po2 = 0;
      drop(r0h_verify_seal(w.data(), w.size(), nullptr, nullptr, fx().seal.data(), fx().seal.size(), &verdict, &po2));
      char* src = nullptr;
      if (good(r0h_circuit_emit_hip(w.data(), w.size(), &src))) r0h_free_error(src);
      { uint32_t root[8]; drop(r0h_control_root_host(w.data(), w.size(), nullptr, nullptr, 4 + (w.empty() ? 0 : w[0] % 3), root)); }  // the blob as a verifier meets it
      break;
    }
    default: {  // a genuine seal with a few words replaced: reaches the deep checks (Merkle paths, FRI) that random words never do
      std::vector<uint32_t> w = fx().seal;
      for (size_t k = 0; k + 8 <= n && !w.empty(); k += 8) {
        uint32_t pos, val;
        memcpy(&pos, d + k, 4); memcpy(&val, d + k + 4, 4);
        w[pos % w.size()] = val;
      }
      if (n % 8 == 1 && !w.empty()) w.resize(w.size() - (d[n - 1] % w.size()));
      int verdict = 0; uint32_t po2 = 0, root[8];
      drop(r0h_verify_seal_bound(fx().blob.data(), fx().blob.size(), nullptr, nullptr, w.data(), w.size(), nullptr, &verdict, &po2, root));
      break;
    }
  }
  return 0;
}
