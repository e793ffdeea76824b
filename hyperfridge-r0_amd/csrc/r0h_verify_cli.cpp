// r0h_verify: check a seal file against a circuit blob on the host -- the compiled counterpart of the reference's
// `verifier verify` step (verifier/src/main.rs:118-126: read the receipt, `receipt.verify(image_id)`, report).
// Needs no GPU.  Exit status: 0 accepted, 1 rejected (reason on stdout), 2 unusable input.
//   usage: r0h_verify <circuit.r0c> <seal.bin>
//          r0h_verify --receipt <receipt.json> <circuit.r0c> --image-id <64 hex> --control-root <po2>:<w0,..,w7> [--control-root ..]
//          r0h_verify --receipt <receipt.json> <circuit.r0c> --image-id <64 hex> --image-circuit <image.r0c>   a trace-circuit receipt that carries
//                     an image proof, checked with the 32 bytes of the image id alone (r0h_receipt_verify_image) -- the reference's call
//          r0h_verify --image-id-of <guest.elf>          prints the image id in the reference's IMAGE_ID.hex form (`host show-image-id`)
//          r0h_verify --receipt <receipt.json> <circuit.r0c> --elf <guest.elf> --control-root ...   (the image id computed from the ELF:
//                     r0h_compute_image_id, what risc0_build embeds as HYPERFRIDGE_ID).  A receipt over the trace circuit binds its
//                     program through the session-wide memory argument, which the verifier completes with the ELF's own words
//                     (r0h_receipt_verify_elf): for such a receipt --elf is the way to an "accepted"; --image-id alone leaves the
//                     session sum unchecked and is reported as not accepted ("program_bound": "no: ...")
//                     `receipt.verify(image_id)` for a composite receipt (r0h_receipt_verify): seals against the control roots, claims
//                     named by the seals, segment chain, journal digest, image id; then the commitment (verifier/src/main.rs:124-128).
//                     Without --control-root the roots are derived from the circuit itself (r0h_control_root_host; a circuit with a
//                     column program); without --image-id / --elf the seals alone are checked and the result is reported as NOT accepted
//                     ("journal_bound": false, exit status 1): valid seals plus any journal would otherwise pass.
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

// verifier/src/main.rs:114-128: read the receipt JSON, verify it against the image id, print the commitment
static int verify_receipt(const char* receipt_path, const char* blob_path, const char* image_hex, std::vector<uint32_t> roots, const char* elf_path = nullptr,
                          const char* image_circuit_path = nullptr) {
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
  uint8_t image_id[32];
  char hex_of_elf[65];
  std::vector<uint8_t> elf;
  if (elf_path && !image_hex) {
    FILE* e = fopen(elf_path, "rb");
    if (!e) { fprintf(stderr, "r0h_verify: cannot open %s\n", elf_path); r0h_receipt_free(rc); return 2; }
    for (size_t got; (got = fread(buf, 1, sizeof buf, e)) > 0;) elf.insert(elf.end(), buf, buf + got);
    fclose(e);
    err = r0h_compute_image_id(elf.data(), elf.size(), image_id);
    if (!err) err = r0h_image_id_to_hex(image_id, hex_of_elf);
    if (err) { fprintf(stderr, "r0h_verify: --elf: %s\n", err); r0h_free_error(err); r0h_receipt_free(rc); return 2; }
    image_hex = hex_of_elf;
  }
  if (image_hex) {
    // eight {:08x} words, each little-endian in the digest: the reference's IMAGE_ID.hex convention (r0h_image_id_from_hex)
    err = r0h_image_id_from_hex(image_hex, image_id);
    if (err) { fprintf(stderr, "r0h_verify: --image-id: %s\n", err); r0h_free_error(err); r0h_receipt_free(rc); return 2; }
  }
  const size_t n_seg = r0h_receipt_n_segments(rc);
  bool derived_roots = false;
  if (image_hex && roots.empty()) {
    // no control roots given: derive them from the circuit itself (r0h_control_root_host: the CODE columns of its column program,
    // committed on the host), one per trace size that occurs in the receipt -- as risc0's verifier looks them up in its own table
    for (size_t i = 0; i < n_seg; i++) {
      const uint32_t* seal; size_t words; uint32_t po2 = 0; int sv = 0;
      err = r0h_receipt_segment(rc, i, &seal, &words, nullptr);
      if (!err) err = r0h_verify_seal(blob.data(), blob.size(), nullptr, nullptr, seal, words, &sv, &po2);
      if (err) { fprintf(stderr, "r0h_verify: %s\n", err); r0h_free_error(err); r0h_receipt_free(rc); return 2; }
      // a root is derived only for a seal that verifies by itself (a bogus seal naming po2 = 24 must not make this verifier commit
      // sixteen million rows before it says no): r0h_receipt_verify reports the seal's own verdict below
      if (sv != R0H_VERIFY_OK) continue;
      bool have = false;
      for (size_t k = 0; k + 9 <= roots.size(); k += 9) have = have || roots[k] == po2;
      if (have || po2 == 0) continue;
      uint32_t root[8];
      err = r0h_control_root_host(blob.data(), blob.size(), nullptr, nullptr, po2, root);
      if (err) { fprintf(stderr, "r0h_verify: no --control-root given and none can be derived: %s\n", err); r0h_free_error(err); roots.clear(); break; }
      roots.push_back(po2);
      roots.insert(roots.end(), root, root + 8);
      derived_roots = true;
    }
  }
  const bool bound = image_hex && !roots.empty();
  int verdict = -1, seal_verdict = R0H_VERIFY_OK;
  size_t at = 0;
  const char* reason = "";
  bool seals_valid = false;
  std::vector<uint32_t> image_blob;  // --image-circuit: the image id alone is enough when the receipt carries an image proof
  if (image_circuit_path && elf.empty() && !read_words(image_circuit_path, &image_blob)) { fprintf(stderr, "r0h_verify: cannot read %s\n", image_circuit_path); r0h_receipt_free(rc); return 2; }
  if (bound) {
    if (!image_blob.empty())
      err = r0h_receipt_verify_image(rc, blob.data(), blob.size(), roots.data(), roots.size() / 9, image_blob.data(), image_blob.size(), nullptr, image_id, &verdict, &at,
                                     &seal_verdict);
    else if (!elf.empty()) err = r0h_receipt_verify_elf(rc, blob.data(), blob.size(), roots.data(), roots.size() / 9, elf.data(), elf.size(), &verdict, &at, &seal_verdict);
    else err = r0h_receipt_verify(rc, blob.data(), blob.size(), roots.data(), roots.size() / 9, image_id, &verdict, &at, &seal_verdict);
    if (err) { fprintf(stderr, "r0h_verify: %s\n", err); r0h_free_error(err); r0h_receipt_free(rc); return 2; }
    reason = verdict == R0H_RECEIPT_V_SEAL ? r0h_verify_reason(seal_verdict) : r0h_receipt_verify_reason(verdict);
    seals_valid = verdict != R0H_RECEIPT_V_SEAL && verdict != R0H_RECEIPT_V_NOT_COMPOSITE && verdict != R0H_RECEIPT_V_NO_CONTROL_ROOT;
  } else {
    // seals only: says nothing about which program ran or which journal it committed to
    seals_valid = r0h_receipt_kind(rc) == R0H_RECEIPT_COMPOSITE && n_seg > 0;
    reason = seals_valid ? "seals checked without image id / control roots: neither the program nor the journal is bound" : "not a composite receipt";
    for (size_t i = 0; i < n_seg && seals_valid; i++) {
      const uint32_t* seal; size_t words;
      err = r0h_receipt_segment(rc, i, &seal, &words, nullptr);
      if (!err) err = r0h_verify_seal(blob.data(), blob.size(), nullptr, nullptr, seal, words, &seal_verdict, nullptr);
      if (err) { fprintf(stderr, "r0h_verify: %s\n", err); r0h_free_error(err); r0h_receipt_free(rc); return 2; }
      if (seal_verdict != R0H_VERIFY_OK) { seals_valid = false; at = i; reason = r0h_verify_reason(seal_verdict); }
    }
  }
  const bool accepted = bound && verdict == R0H_RECEIPT_V_OK;
  const uint8_t* journal; size_t jn, off = 0, len = 0;
  (void)r0h_receipt_journal(rc, &journal, &jn);
  err = r0h_journal_commitment_span(journal, jn, &off, &len);
  if (err) { r0h_free_error(err); len = 0; }
  // what ties the receipt to a program: the session sum completed with the ELF's own words (trace circuit), or -- synthetic circuits --
  // only the chain of claims from the image id (their seals prove that a satisfying trace naming the claim exists, not that the program ran)
  const char* program_bound = !accepted ? (verdict == R0H_RECEIPT_V_NEEDS_IMAGE ? "no: the receipt binds its program through the session sum; give --elf, or --image-circuit if it carries an image proof" : "no")
                                        : !image_blob.empty() ? "yes: image id and the session sum balanced with the receipt's image proof (whose in-circuit digest is the root the image id names)"
                                        : !elf.empty() ? "yes: image id and, for a trace-circuit receipt, the session sum over the ELF's image words" : "by the claims' chain from the image id";
  printf("{\"accepted\": %s, \"seals_valid\": %s, \"journal_bound\": %s, \"program_bound\": \"%s\", \"segments\": %zu, \"segment_at_fault\": %zu, \"reason\": \"%s\", \"control_roots\": \"%s\", \"commitment\": ",
         accepted ? "true" : "false", seals_valid ? "true" : "false", accepted ? "true" : "false", program_bound, n_seg, at, accepted ? "ok" : reason,
         derived_roots ? "derived from the circuit" : roots.empty() ? "none" : "given");
  print_json_string(journal + off, len);
  printf("}\n");
  r0h_receipt_free(rc);
  return accepted ? 0 : 1;
}

int main(int argc, char** argv) {
  if (argc == 3 && !strcmp(argv[1], "--image-id-of")) {  // `host show-image-id` (host/src/main.rs): the id a verifier holds a receipt against
    FILE* e = fopen(argv[2], "rb");
    if (!e) { fprintf(stderr, "r0h_verify: cannot open %s\n", argv[2]); return 2; }
    std::vector<uint8_t> elf;
    char buf[65536];
    for (size_t got; (got = fread(buf, 1, sizeof buf, e)) > 0;) elf.insert(elf.end(), buf, buf + got);
    fclose(e);
    uint8_t id[32];
    char hex[65];
    const char* err = r0h_compute_image_id(elf.data(), elf.size(), id);
    if (!err) err = r0h_image_id_to_hex(id, hex);
    if (err) { fprintf(stderr, "r0h_verify: %s\n", err); r0h_free_error(err); return 2; }
    printf("%s\n", hex);
    return 0;
  }
  if (argc >= 4 && !strcmp(argv[1], "--receipt")) {
    const char* image_hex = nullptr;
    const char* elf_path = nullptr;
    const char* image_circuit_path = nullptr;
    std::vector<uint32_t> roots;  // records of [po2, root[8]]
    for (int i = 4; i + 1 < argc; i += 2) {
      if (!strcmp(argv[i], "--image-id")) image_hex = argv[i + 1];
      else if (!strcmp(argv[i], "--elf")) elf_path = argv[i + 1];
      else if (!strcmp(argv[i], "--image-circuit")) image_circuit_path = argv[i + 1];
      else if (!strcmp(argv[i], "--control-root")) {
        unsigned v[9];
        if (sscanf(argv[i + 1], "%u:%u,%u,%u,%u,%u,%u,%u,%u", &v[0], &v[1], &v[2], &v[3], &v[4], &v[5], &v[6], &v[7], &v[8]) != 9) {
          fprintf(stderr, "r0h_verify: --control-root wants <po2>:<w0,..,w7>\n");
          return 2;
        }
        roots.insert(roots.end(), v, v + 9);
      } else { fprintf(stderr, "r0h_verify: unknown option %s\n", argv[i]); return 2; }
    }
    return verify_receipt(argv[2], argv[3], image_hex, roots, elf_path, image_circuit_path);
  }
  if (argc != 3) {
    printf("usage: r0h_verify <circuit.r0c> <seal.bin>\n       r0h_verify --receipt <receipt.json> <circuit.r0c> --image-id <64 hex> --control-root <po2>:<w0,..,w7>\n%s\n", r0h_version());
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
