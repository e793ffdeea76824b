// Host side of the log-derivative argument (blob section LOGUP): the multiplicity columns of a witness that lives in host memory --
// what the tests' reference witness (r0h_vm_trace_witness) is completed with, and what a caller that builds its own DATA group needs
// before committing it.  The device form is csrc/logup.hip; this one is plain loops over the same description.
#include <string.h>

#include <vector>

#include "../../include/r0hip_circuit.h"
#include "circuit.hpp"

using namespace r0h;

extern "C" {

const char* r0h_logup_multiplicities_host(const uint32_t* blob, size_t blob_words, uint32_t po2, uint32_t* data, const uint32_t* global) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(blob && data, "r0h_logup_multiplicities_host: NULL argument");
  r0h_circuit c;
  R0H_TRY(parse_blob(&c, blob, blob_words));
  if (c.logup.tables.empty()) return nullptr;
  R0H_REQUIRE(po2 >= 16 && po2 <= R0H_MAX_PO2, "r0h_logup_multiplicities_host: the tables have 2^16 rows: po2 %u outside [16, %u]", po2, (unsigned)R0H_MAX_PO2);
  const size_t n = (size_t)1 << po2;
  std::vector<std::vector<uint32_t>> hist(c.logup.tables.size(), std::vector<uint32_t>(65536, 0));
  auto form = [&](const Lf& lf, size_t r) {
    uint32_t acc = 0;
    for (const LfTerm& t : lf.terms) {
      uint32_t v = enc(t.coef);
      if (t.global) v = mul(v, global ? global[t.global - 1] : 0u);
      if (t.col) {
        const uint32_t ref = t.col - 1;
        if ((ref >> 28) != R0H_GROUP_DATA) return 0xffffffffu;  // a lookup's value reads DATA only
        v = mul(v, data[(size_t)(ref & 0xfffffu) * n + r]);
      }
      acc = add(acc, v);
    }
    return acc;
  };
  for (uint32_t j = 0; j < c.logup.n_chain; j++)
    for (const LogupFraction& f : c.logup.accs[j].fr) {
      if (!f.table || f.table > hist.size()) continue;
      for (size_t r = 0; r < n; r++) {
        if (form(f.num, r) != ONE) continue;
        const uint32_t raw = form(f.parts[1].lf, r);
        R0H_REQUIRE(raw != 0xffffffffu, "r0h_logup_multiplicities_host: a lookup's value reads another group than DATA");
        uint32_t v = dec(neg(raw));
        if (f.table == R0H_TABLE_AND) {
          v -= R0H_TAG_AND;
          R0H_REQUIRE(!(v >> 24) && ((v & 255u) & ((v >> 8) & 255u)) == v >> 16, "r0h_logup_multiplicities_host: row %zu looks up a value that is not in the byte-AND table", r);
          v &= 0xffffu;
        } else {
          R0H_REQUIRE(!(v >> 16), "r0h_logup_multiplicities_host: row %zu looks up %u in the 16-bit range table", r, v);
        }
        hist[f.table - 1][v]++;
      }
    }
  for (size_t k = 0; k < c.logup.tables.size(); k++) {
    uint32_t* col = data + (size_t)c.logup.tables[k].data_col * n;
    memset(col, 0, n * 4);
    for (uint32_t v = 0; v < 65536; v++) col[v] = enc(hist[k][v]);
  }
  return nullptr;
  R0H_GUARD_END
}

}  // extern "C"
