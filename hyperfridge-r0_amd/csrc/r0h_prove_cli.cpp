// r0h_prove: a minimal compiled host over the C ABI (no Python, no torch): load a circuit blob, generate a synthetic
// witness on the device, prove K segments, print throughput and a digest of the seal.
// It stands where hyperfridge's `host prove-camt53` stands relative to risc0 (host/src/main.rs:420-423 obtains a prover and
// calls prove once); everything risc0-specific above the segment prover (executor, receipts) is out of scope.
//   usage: r0h_prove <circuit.r0c> [--code-object file.hsaco] [--po2 N] [--segments K] [--seed S] [--device D] [--seal-out file]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/r0hip.h"

static void die(const char* what, const char* err) {
  fprintf(stderr, "r0h_prove: %s: %s\n", what, err);
  r0h_free_error(err);
  exit(2);
}
#define CHECK(call)                      \
  do {                                   \
    const char* e__ = (call);            \
    if (e__) die(#call, e__);            \
  } while (0)

int main(int argc, char** argv) {
  if (argc < 2 || !strcmp(argv[1], "--help") || !strcmp(argv[1], "-h")) {
    printf("usage: r0h_prove <circuit.r0c> [--code-object file.hsaco] [--po2 N] [--segments K] [--seed S] [--device D] [--seal-out file]\n%s\n", r0h_version());
    return argc < 2 ? 1 : 0;
  }
  std::string blob_path = argv[1], co_path, seal_out;
  unsigned po2 = 16, segments = 1, device = 0;
  unsigned long long seed = 1;
  for (int i = 2; i + 1 < argc; i += 2) {
    if (!strcmp(argv[i], "--code-object")) co_path = argv[i + 1];
    else if (!strcmp(argv[i], "--po2")) po2 = (unsigned)atoi(argv[i + 1]);
    else if (!strcmp(argv[i], "--segments")) segments = (unsigned)atoi(argv[i + 1]);
    else if (!strcmp(argv[i], "--seed")) seed = strtoull(argv[i + 1], nullptr, 10);
    else if (!strcmp(argv[i], "--device")) device = (unsigned)atoi(argv[i + 1]);
    else if (!strcmp(argv[i], "--seal-out")) seal_out = argv[i + 1];
    else { fprintf(stderr, "r0h_prove: unknown option %s\n", argv[i]); return 1; }
  }
  FILE* f = fopen(blob_path.c_str(), "rb");
  if (!f) { fprintf(stderr, "r0h_prove: cannot open %s\n", blob_path.c_str()); return 1; }
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  std::vector<uint32_t> blob((size_t)sz / 4);
  if (fread(blob.data(), 4, blob.size(), f) != blob.size()) { fprintf(stderr, "r0h_prove: short read\n"); return 1; }
  fclose(f);

  r0h_ctx* ctx = nullptr;
  CHECK(r0h_ctx_create((int)device, &ctx));
  r0h_circuit* circ = nullptr;
  CHECK(r0h_circuit_load(ctx, blob.data(), blob.size(), co_path.empty() ? nullptr : co_path.c_str(), &circ));
  const size_t n = (size_t)1 << po2;
  r0h_buf *code = nullptr, *data = nullptr;
  CHECK(r0h_buf_alloc(ctx, (size_t)r0h_circuit_group_size(circ, R0H_GROUP_CODE) * n * 4, &code));
  CHECK(r0h_buf_alloc(ctx, (size_t)r0h_circuit_group_size(circ, R0H_GROUP_DATA) * n * 4, &data));
  std::vector<uint32_t> global(r0h_circuit_n_global(circ) + 1), seal((size_t)1 << 20);
  size_t words = 0;
  double total = 0;
  for (unsigned s = 0; s < segments; s++) {
    CHECK(r0h_witgen(ctx, circ, po2, seed + s, code, data, global.data()));
    auto t0 = std::chrono::steady_clock::now();
    CHECK(r0h_prove_segment(ctx, circ, po2, code, data, global.data(), seal.data(), seal.size(), &words));
    total += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  uint64_t h = 1469598103934665603ull;  // FNV-1a over the last seal, to compare runs
  for (size_t i = 0; i < words; i++) { h ^= seal[i]; h *= 1099511628211ull; }
  printf("{\"segments\": %u, \"po2\": %u, \"seal_words\": %zu, \"seal_fnv1a\": \"%016llx\", \"seconds\": %.6f, \"segments_per_s\": %.4f}\n", segments, po2, words,
         (unsigned long long)h, total, segments / total);
  if (!seal_out.empty()) {
    FILE* o = fopen(seal_out.c_str(), "wb");
    if (!o || fwrite(seal.data(), 4, words, o) != words) { fprintf(stderr, "r0h_prove: cannot write %s\n", seal_out.c_str()); return 1; }
    fclose(o);
  }
  CHECK(r0h_buf_free(code));
  CHECK(r0h_buf_free(data));
  CHECK(r0h_circuit_free(circ));
  CHECK(r0h_ctx_destroy(ctx));
  return 0;
}
