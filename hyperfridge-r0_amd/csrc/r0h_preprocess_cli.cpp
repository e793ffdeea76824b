// r0h_preprocess: the compiled counterpart of data/checkResponse.sh (the step host/src/main.rs:143-151 runs as a child process
// before proving): cut the guest inputs out of an EBICS response, run the script's checks, write the `<xml>-*` files `host` then
// reads (host/src/main.rs:206-227).  Host only -- no GPU, no openssl / xmllint / perl / zlib-flate.
//   usage: r0h_preprocess <response.xml> --pub-bank bank.pem --pub-client client.pem --pub-witness witness.pem
//                         (--client-key client_private.pem | --tx-key-raw <xml>-TransactionKeyDecrypt.bin)
//                         (--witness-key witness_private.pem | --witness-hex <xml>-Witness.hex) [--out-dir dir]
// The two steps that need a PRIVATE key -- decrypting the transaction key and signing as the witness (`openssl pkeyutl -decrypt` /
// `-sign`, checkResponse.sh:231-236, 276-279) -- are done here when the key is given, or their results are taken as files.
// Exit status: 0 all checks passed, 1 a check failed (named on stdout), 2 unusable input.
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/r0hip.h"

static bool slurp(const std::string& path, std::string* out) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  char buf[65536];
  for (size_t got; (got = fread(buf, 1, sizeof buf, f)) > 0;) out->append(buf, got);
  fclose(f);
  return true;
}
static bool spill(const std::string& path, const uint8_t* p, size_t n) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) return false;
  const bool ok = fwrite(p, 1, n, f) == n;
  fclose(f);
  return ok;
}
#define CHECK(call)                                                       \
  do {                                                                    \
    const char* e__ = (call);                                             \
    if (e__) { fprintf(stderr, "r0h_preprocess: %s\n", e__); r0h_free_error(e__); return 2; } \
  } while (0)

int main(int argc, char** argv) {
  if (argc < 2 || !strcmp(argv[1], "--help")) {
    printf("usage: r0h_preprocess <response.xml> --pub-bank f --pub-client f --pub-witness f (--client-key f | --tx-key-raw f) (--witness-key f | --witness-hex f) [--out-dir d]\n%s\n", r0h_version());
    return argc < 2 ? 2 : 0;
  }
  std::string xml_path = argv[1], pub_bank, pub_client, pub_witness, tx_raw, witness_hex, out_dir, client_key, witness_key;
  for (int i = 2; i + 1 < argc; i += 2) {
    std::string* dst = !strcmp(argv[i], "--pub-bank") ? &pub_bank : !strcmp(argv[i], "--pub-client") ? &pub_client : !strcmp(argv[i], "--pub-witness") ? &pub_witness
                       : !strcmp(argv[i], "--tx-key-raw") ? &tx_raw : !strcmp(argv[i], "--witness-hex") ? &witness_hex : !strcmp(argv[i], "--client-key") ? &client_key
                       : !strcmp(argv[i], "--witness-key") ? &witness_key : nullptr;
    if (!strcmp(argv[i], "--out-dir")) { out_dir = argv[i + 1]; continue; }
    if (!dst) { fprintf(stderr, "r0h_preprocess: unknown option %s\n", argv[i]); return 2; }
    if (!slurp(argv[i + 1], dst)) { fprintf(stderr, "r0h_preprocess: cannot read %s\n", argv[i + 1]); return 2; }
  }
  std::string xml;
  if (!slurp(xml_path, &xml)) { fprintf(stderr, "r0h_preprocess: cannot read %s\n", xml_path.c_str()); return 2; }
  if (pub_bank.empty() || pub_client.empty() || pub_witness.empty() || (tx_raw.empty() && client_key.empty()) || (witness_hex.empty() && witness_key.empty())) {
    fprintf(stderr, "r0h_preprocess: --pub-bank, --pub-client, --pub-witness, (--client-key | --tx-key-raw) and (--witness-key | --witness-hex) are all needed\n");
    return 2;
  }
  r0h_ebics* e = nullptr;
  CHECK(r0h_ebics_parse(xml.data(), xml.size(), &e));
  if (!client_key.empty()) {  // checkResponse.sh:231-236
    uint8_t raw[1024], k[16];
    size_t n = 0;
    int ok = 0;
    CHECK(r0h_ebics_decrypt_transaction_key(e, client_key.data(), client_key.size(), raw, sizeof raw, &n, k, &ok));
    tx_raw.assign((const char*)raw, n);  // re-checked against the public key below, like a block read from a file
  }
  if (!witness_key.empty()) {  // checkResponse.sh:276-279
    char* hex = nullptr;
    CHECK(r0h_ebics_witness_sign(e, witness_key.data(), witness_key.size(), &hex));
    witness_hex = hex;
    r0h_free_error(hex);
  }
  int digest = 0, bank = 0, txk = 0, wit = 0;
  uint8_t key[16];
  CHECK(r0h_ebics_check_digest(e, &digest));
  CHECK(r0h_ebics_verify_bank_signature(e, pub_bank.data(), pub_bank.size(), &bank));
  CHECK(r0h_ebics_check_transaction_key(e, pub_client.data(), pub_client.size(), (const uint8_t*)tx_raw.data(), tx_raw.size(), key, &txk));
  CHECK(r0h_ebics_verify_witness(e, pub_witness.data(), pub_witness.size(), witness_hex.data(), witness_hex.size(), &wit));
  size_t n_docs = 0;
  std::string decrypt_error;
  if (txk) {
    const char* err = r0h_ebics_decrypt_order_data(e, key);
    if (err) { decrypt_error = err; r0h_free_error(err); } else n_docs = r0h_ebics_n_documents(e);
  }
  if (!out_dir.empty()) {  // the files host/src/main.rs:206-227 reads, named as the script names them
    std::string stem = xml_path.substr(xml_path.find_last_of('/') == std::string::npos ? 0 : xml_path.find_last_of('/') + 1);
    static const struct { int which; const char* suffix; } parts[] = {{R0H_EBICS_AUTHENTICATED, "-authenticated"}, {R0H_EBICS_SIGNED_INFO, "-SignedInfo"},
                                                                      {R0H_EBICS_SIGNATURE_VALUE, "-SignatureValue"}, {R0H_EBICS_ORDER_DATA, "-OrderData"}};
    for (const auto& p : parts) {
      const uint8_t* b; size_t n;
      CHECK(r0h_ebics_part(e, p.which, &b, &n));
      if (!spill(out_dir + "/" + stem + p.suffix, b, n)) { fprintf(stderr, "r0h_preprocess: cannot write into %s\n", out_dir.c_str()); return 2; }
    }
    spill(out_dir + "/" + stem + "-TransactionKeyDecrypt.bin", (const uint8_t*)tx_raw.data(), tx_raw.size());
    spill(out_dir + "/" + stem + "-Witness.hex", (const uint8_t*)witness_hex.data(), witness_hex.size());
    for (size_t i = 0; i < n_docs; i++) {
      const char* name; const uint8_t* d; size_t n;
      CHECK(r0h_ebics_document(e, i, &name, &d, &n));
      std::string base = name;
      for (char& c : base) if (c == '/' || c == '\\') c = '_';
      spill(out_dir + "/" + stem + "-camt53-" + base, d, n);
    }
  }
  const bool ok = digest && bank && txk && wit && n_docs > 0;
  printf("{\"ok\": %s, \"digest\": %s, \"bank_signature\": %s, \"transaction_key\": %s, \"witness_signature\": %s, \"documents\": %zu%s%s%s}\n", ok ? "true" : "false",
         digest ? "true" : "false", bank ? "true" : "false", txk ? "true" : "false", wit ? "true" : "false", n_docs, decrypt_error.empty() ? "" : ", \"decrypt_error\": \"",
         decrypt_error.c_str(), decrypt_error.empty() ? "" : "\"");
  r0h_ebics_free(e);
  return ok ? 0 : 1;
}
