// r0h_verify: check a seal file against a circuit blob on the host -- the compiled counterpart of the reference's
// `verifier verify` step (verifier/src/main.rs:118-126: read the receipt, `receipt.verify(image_id)`, report).
// Needs no GPU.  Exit status: 0 accepted, 1 rejected (reason on stdout), 2 unusable input.
//   usage: r0h_verify <circuit.r0c> <seal.bin>
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

int main(int argc, char** argv) {
  if (argc != 3) {
    printf("usage: r0h_verify <circuit.r0c> <seal.bin>\n%s\n", r0h_version());
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
