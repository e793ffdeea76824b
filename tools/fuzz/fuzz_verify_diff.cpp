// Differential fuzzing of the two verifiers: the product's (csrc/verify.cpp, what `receipt.verify` rests on) and the CPU oracle's
// (oracle/orc_prove.c) must return the same verdict for every edit of a genuine seal -- accept together, and name the same first
// failed check when they reject.  Built and run by tools/fuzz/run_diff.sh (CPU only, libFuzzer + ASan).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/r0hip.h"
extern "C" {
#include "../../oracle/orc.h"
}

static std::vector<uint8_t> slurp(const std::string& path) {
  std::vector<uint8_t> v;
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) { fprintf(stderr, "fuzz_verify_diff: cannot read %s (set R0H_FUZZ_ROOT)\n", path.c_str()); abort(); }
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  v.resize(n > 0 ? (size_t)n : 0);
  if (n > 0 && fread(v.data(), 1, v.size(), f) != v.size()) abort();
  fclose(f);
  return v;
}

struct Fx {
  std::vector<uint32_t> blob, seal;
  orc_circuit_t* oc = nullptr;
  Fx() {
    const char* env = getenv("R0H_FUZZ_ROOT");
    const std::string root = env ? env : ".";
    std::vector<uint8_t> c = slurp(root + "/circuits/tiny.r0c"), s = slurp(root + "/tests/golden/seal_tiny_po2_9_seed_1.npy");
    blob.resize(c.size() / 4);
    memcpy(blob.data(), c.data(), blob.size() * 4);
    const size_t off = 10 + (s[8] | (size_t)s[9] << 8);
    seal.resize((s.size() - off) / 4);
    memcpy(seal.data(), s.data() + off, seal.size() * 4);
    oc = orc_circuit_parse(blob.data(), blob.size());
    if (!oc) abort();
    int v = -1; uint32_t po2 = 0;
    const char* e = r0h_verify_seal(blob.data(), blob.size(), nullptr, nullptr, seal.data(), seal.size(), &v, &po2);
    if (e || v != 0 || orc_verify_segment(oc, blob.data(), blob.size(), seal.data(), seal.size()) != 0) { fprintf(stderr, "the genuine seal must verify\n"); abort(); }
  }
};
static Fx& fx() { static Fx f; return f; }

extern "C" int LLVMFuzzerTestOneInput(const uint8_t* d, size_t n) {
  std::vector<uint32_t> w = fx().seal;
  // records of 9 bytes: op, position, value.  op 0: replace a word; 1: add to it; 2: cut the seal there; 3: append the value;
  // 4: copy a word from elsewhere (position taken from the value)
  for (size_t k = 0; k + 9 <= n && !w.empty(); k += 9) {
    uint32_t pos, val;
    memcpy(&pos, d + k + 1, 4);
    memcpy(&val, d + k + 5, 4);
    switch (d[k] % 5) {
      case 0: w[pos % w.size()] = val; break;
      case 1: w[pos % w.size()] += val; break;
      case 2: w.resize(pos % (w.size() + 1)); break;
      case 3: if (w.size() < fx().seal.size() + 64) w.push_back(val); break;
      default: w[pos % w.size()] = w[val % w.size()]; break;
    }
  }
  int got = -1; uint32_t po2 = 0;
  const char* e = r0h_verify_seal(fx().blob.data(), fx().blob.size(), nullptr, nullptr, w.data(), w.size(), &got, &po2);
  if (e) { fprintf(stderr, "product verifier returned an error instead of a verdict: %s\n", e); abort(); }
  const int want = orc_verify_segment(fx().oc, fx().blob.data(), fx().blob.size(), w.data(), w.size());
  if (got != want) {
    fprintf(stderr, "verdicts differ: product %d (%s), oracle %d (%s); %zu words\n", got, r0h_verify_reason(got), want, orc_verify_strerror(want), w.size());
    abort();
  }
  return 0;
}
