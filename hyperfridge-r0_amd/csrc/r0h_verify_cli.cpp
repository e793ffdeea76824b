// r0h_verify: check a seal file against a circuit blob on the host -- the compiled counterpart of the reference's
// `verifier verify` step (verifier/src/main.rs:118-126: read the receipt, `receipt.verify(image_id)`, report).
// Needs no GPU.  Exit status: 0 accepted, 1 rejected (reason on stdout), 2 unusable input.
//   usage: r0h_verify <circuit.r0c> <seal.bin>
//          r0h_verify --receipt <receipt.json> <circuit.r0c>     every segment seal of a Receipt JSON, then the commitment
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "../../include/r0hip.h"

static bool read_words(const char* path, std::vector<uint32_t>* out) {
  FILE* f = fopen(path, "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  out->resize(sz > 0 ? (size_t)sz / 4 : 0);
  bool ok = sz >= 0 && sz % 4 == 0 && fread(out->data(), 4, out->size(), f) == out->size();
  fclose(f);
  return ok;
}

static void print_json_string(const uint8_t* p, size_t n) {
  putchar('"');
  for (size_t i = 0; i < n; i++) {
    unsigned char c = p[i];
    if (c == '"' || c == '\\') { putchar('\\'); putchar(c); }
    else if (c < 0x20) printf("\\u%04x", c);
    else putchar(c);
  }
  putchar('"');
}

// verifier/src/main.rs:114-126: read the receipt JSON, verify it, print the commitment
static int verify_receipt(const char* receipt_path, const char* blob_path) {
  std::vector<uint32_t> blob;
  if (!read_words(blob_path, &blob)) { fprintf(stderr, "r0h_verify: cannot read %s as 32-bit words\n", blob_path); return 2; }
  FILE* f = fopen(receipt_path, "rb");
  if (!f) { fprintf(stderr, "r0h_verify: cannot open %s\n", receipt_path); return 2; }
  std::vector<char> text;
  char buf[65536];
  for (size_t got; (got = fread(buf, 1, sizeof buf, f)) > 0;) text.insert(text.end(), buf, buf + got);
  fclose(f);
  r0h_receipt* rc = nullptr;
  const char* err = r0h_receipt_parse(text.data(), text.size(), &rc);
  if (err) { fprintf(stderr, "r0h_verify: %s\n", err); r0h_free_error(err); return 2; }
  const size_t n_seg = r0h_receipt_n_segments(rc);
  int verdict = R0H_VERIFY_OK;
  const char* reason = "ok";
  if (r0h_receipt_kind(rc) == R0H_RECEIPT_FAKE) {
    verdict = -1;
    reason = "Fake receipt (dev mode): nothing to verify";
  } else if (n_seg == 0) {
    verdict = -1;
    reason = "composite receipt without segments";
  }
  for (size_t i = 0; i < n_seg && verdict == R0H_VERIFY_OK; i++) {
    const uint32_t* seal; size_t words;
    err = r0h_receipt_segment(rc, i, &seal, &words, nullptr);
    if (!err) err = r0h_verify_seal(blob.data(), blob.size(), nullptr, nullptr, seal, words, &verdict, nullptr);
    if (err) { fprintf(stderr, "r0h_verify: %s\n", err); r0h_free_error(err); r0h_receipt_free(rc); return 2; }
    reason = r0h_verify_reason(verdict);
  }
  const uint8_t* journal; size_t jn, off = 0, len = 0;
  (void)r0h_receipt_journal(rc, &journal, &jn);
  err = r0h_journal_commitment_span(journal, jn, &off, &len);
  if (err) { r0h_free_error(err); len = 0; }
  printf("{\"accepted\": %s, \"segments\": %zu, \"reason\": \"%s\", \"commitment\": ", verdict == R0H_VERIFY_OK ? "true" : "false", n_seg, reason);
  print_json_string(journal + off, len);
  printf("}\n");
  r0h_receipt_free(rc);
  return verdict == R0H_VERIFY_OK ? 0 : 1;
}

int main(int argc, char** argv) {
  if (argc == 4 && !strcmp(argv[1], "--receipt")) return verify_receipt(argv[2], argv[3]);
  if (argc != 3) {
    printf("usage: r0h_verify <circuit.r0c> <seal.bin>\n       r0h_verify --receipt <receipt.json> <circuit.r0c>\n%s\n", r0h_version());
    return 2;
  }
  std::vector<uint32_t> blob, seal;
  if (!read_words(argv[1], &blob)) { fprintf(stderr, "r0h_verify: cannot read %s as 32-bit words\n", argv[1]); return 2; }
  if (!read_words(argv[2], &seal)) { fprintf(stderr, "r0h_verify: cannot read %s as 32-bit words\n", argv[2]); return 2; }
  int verdict = -1;
  uint32_t po2 = 0;
  const char* err = r0h_verify_seal(blob.data(), blob.size(), nullptr, nullptr, seal.data(), seal.size(), &verdict, &po2);
  if (err) {
    fprintf(stderr, "r0h_verify: %s\n", err);
    r0h_free_error(err);
    return 2;
  }
  printf("{\"accepted\": %s, \"verdict\": %d, \"reason\": \"%s\", \"po2\": %u, \"seal_words\": %zu}\n", verdict == R0H_VERIFY_OK ? "true" : "false",
         verdict, r0h_verify_reason(verdict), po2, seal.size());
  return verdict == R0H_VERIFY_OK ? 0 : 1;
}
