// Circuit side of the prover: blob loading, eval_check code generation + launch, synthetic witness generation and
// accumulation.  Replaces the CircuitHal half of risc0-circuit-rv32im 4.0.4 / risc0-circuit-rv32im-sys 4.0.2
// (`eval_check` + generated `poly_fp`, `generate_witness`, `step_accum`) -- SURVEY.md 8(a) a9-a11.
//
// Upstream compiles a machine-generated constraint polynomial into its kernels at build time.  Here the circuit is
// data (include/r0hip_circuit.h): r0h_circuit_emit_hip turns its PolyExtStep program into straight-line HIP --
//   check[i] = (sum_t poly_mix^{p_t} * val_t(i)) / ((3 w^i)^N - 1)      on the 4N coset domain
// where AndCond gates are flattened into their inner terms (val = cond * inner value, power = outer + inner), the
// powers of poly_mix are wave-uniform (scalar loads), every lane owns one domain point and reads tap (col, back) at
// row (i - 4*back) mod 4N: consecutive lanes hit consecutive words of one column.  The program is cut into several
// kernels so that each has a bounded live set; partial sums accumulate through the check buffer.
#include <hip/hiprtc.h>

#include <map>
#include <memory>
#include <sstream>

#include "../../include/r0hip_circuit.h"
#include "circuit.hpp"

namespace r0h {

const char* parse_blob(r0h_circuit* c, const uint32_t* w, size_t n_words) {
  R0H_REQUIRE(n_words >= 3 && w[0] == R0H_BLOB_MAGIC && w[1] == 1, "circuit blob: bad magic or version");
  size_t pos = 3;
  bool seen[16] = {false};
  for (uint32_t s = 0; s < w[2]; s++) {
    R0H_REQUIRE(pos + 2 <= n_words, "circuit blob: truncated section header");
    uint32_t tag = w[pos], len = w[pos + 1];
    const uint32_t* p = w + pos + 2;
    R0H_REQUIRE(pos + 2 + len <= n_words, "circuit blob: section %u overruns the blob", tag);
    if (tag < 16) seen[tag] = true;
    switch (tag) {
      case R0H_SEC_GROUPS:
        R0H_REQUIRE(len >= 3, "circuit blob: GROUPS too short");
        memcpy(c->group_size, p, 12);
        break;
      case R0H_SEC_TAPS: {
        R0H_REQUIRE(len >= 1 && len == 1 + 3 * (size_t)p[0], "circuit blob: TAPS length mismatch");
        c->taps.resize(p[0]);
        if (p[0]) memcpy(c->taps.data(), p + 1, 12 * (size_t)p[0]);
        break;
      }
      case R0H_SEC_GLOBALS:
        R0H_REQUIRE(len >= 2 && (len == 2 + (size_t)p[0] || len == 2), "circuit blob: GLOBALS length mismatch");
        if (len == 2) { c->n_global = p[0]; c->n_mix = p[1]; c->global_cols.assign(p[0], 0); break; }
        c->n_global = p[0]; c->n_mix = p[1];
        c->global_cols.assign(p + 2, p + 2 + p[0]);
        break;
      case R0H_SEC_POLY:
        R0H_REQUIRE(len >= 2 && len == 2 + 4 * (size_t)p[0], "circuit blob: POLY length mismatch");
        c->ret = p[1];
        c->steps.resize(p[0]);
        if (p[0]) memcpy(c->steps.data(), p + 2, 16 * (size_t)p[0]);
        break;
      case R0H_SEC_WITGEN: {
        R0H_REQUIRE(len >= 1 && len >= 2 + 2 * (size_t)p[0], "circuit blob: WITGEN too short");
        c->code_cols.resize(p[0]);
        if (p[0]) memcpy(c->code_cols.data(), p + 1, 8 * (size_t)p[0]);
        const uint32_t* q = p + 1 + 2 * (size_t)p[0];
        R0H_REQUIRE(len == 2 + 2 * (size_t)p[0] + 5 * (size_t)q[0], "circuit blob: WITGEN length mismatch");
        c->data_cols.resize(q[0]);
        if (q[0]) memcpy(c->data_cols.data(), q + 1, 20 * (size_t)q[0]);
        break;
      }
      case R0H_SEC_ACCUM:
        R0H_REQUIRE(len >= 1 && len == 1 + 3 * (size_t)p[0], "circuit blob: ACCUM length mismatch");
        c->acc_cols.resize(p[0]);
        if (p[0]) memcpy(c->acc_cols.data(), p + 1, 12 * (size_t)p[0]);
        break;
      case R0H_SEC_ACCUM_FP:
        R0H_REQUIRE(len >= 1 && len == 1 + 13 * (size_t)p[0], "circuit blob: ACCUM_FP length mismatch");
        c->acc_fp.resize(p[0]);
        if (p[0]) memcpy(c->acc_fp.data(), p + 1, 52 * (size_t)p[0]);
        break;
      case R0H_SEC_INFO:
        R0H_REQUIRE(len == 4, "circuit blob: INFO must be 4 words");
        memcpy(c->info, p, 16);
        break;
      case R0H_SEC_LATE:
        R0H_REQUIRE(len == 1, "circuit blob: LATE must be 1 word");
        c->n_late = p[0];
        break;
      case R0H_SEC_PERIODIC:
        R0H_REQUIRE(len >= 2 && p[0] && (uint64_t)p[0] * p[1] + 2 == len, "circuit blob: PERIODIC length mismatch");
        c->period = p[0];
        c->periodic.assign(p + 2, p + len);
        for (uint32_t v : c->periodic) R0H_REQUIRE(v < P, "circuit blob: PERIODIC value is not a canonical field word");
        break;
      case R0H_SEC_SPONGE:
        R0H_REQUIRE(len == 3, "circuit blob: SPONGE must be 3 words");
        c->has_sponge = true; c->sponge_code = p[0]; c->sponge_data = p[1]; c->sponge_global = p[2];
        break;
      case R0H_SEC_LOGUP: {
        size_t at = 0;
        auto word = [&](uint32_t* out) -> bool { if (at >= len) return false; *out = p[at++]; return true; };
        auto form = [&](Lf* lf) -> bool {
          uint32_t n;
          if (!word(&n) || n > 64) return false;
          lf->terms.resize(n);
          for (LfTerm& t : lf->terms)
            if (!word(&t.coef) || !word(&t.global) || !word(&t.col) || t.coef >= P) return false;
          return true;
        };
        uint32_t n_acc = 0, n_tab = 0;
        R0H_REQUIRE(word(&n_acc) && word(&n_tab) && n_acc >= 1 && n_acc <= 64 && n_tab <= 8, "circuit blob: LOGUP header");
        c->logup.tables.resize(n_tab);
        for (LogupTable& t : c->logup.tables) R0H_REQUIRE(word(&t.data_col) && word(&t.kind) && (t.kind == R0H_TABLE_R16 || t.kind == R0H_TABLE_AND), "circuit blob: LOGUP table");
        c->logup.accs.resize(n_acc);
        bool chain_over = false;
        for (LogupAcc& a : c->logup.accs) {
          uint32_t nf = 0;
          R0H_REQUIRE(word(&nf) && word(&a.final_global) && nf == 4, "circuit blob: a LOGUP accumulator has four fractions");
          if (a.final_global == 0xffffffffu) { R0H_REQUIRE(!chain_over, "circuit blob: LOGUP chain links come first"); c->logup.n_chain++; }
          else chain_over = true;
          a.fr.resize(nf);
          for (LogupFraction& f : a.fr) {
            uint32_t np = 0;
            R0H_REQUIRE(word(&f.table) && f.table <= 2 && form(&f.num) && word(&np) && np >= 1 && np <= 8, "circuit blob: LOGUP fraction");
            f.parts.resize(np);
            for (LogupPart& q : f.parts) R0H_REQUIRE(word(&q.ch_kind) && q.ch_kind <= 2 && word(&q.ch_idx) && form(&q.lf), "circuit blob: LOGUP part");
            if (f.table) R0H_REQUIRE(np == 2 && f.parts[1].ch_kind == 0, "circuit blob: a lookup's value is its second part");
          }
        }
        R0H_REQUIRE(at == len, "circuit blob: LOGUP length mismatch");
        break;
      }
      default: break;
    }
    pos += 2 + len;
  }
  for (int t = 1; t <= 4; t++) R0H_REQUIRE(seen[t], "circuit blob: section %d missing", t);
  // WITGEN + ACCUM (the synthetic column program) are optional: a circuit imported from risc0 brings its own witness
  const bool any_accum = seen[R0H_SEC_ACCUM] || seen[R0H_SEC_ACCUM_FP] || seen[R0H_SEC_LOGUP];
  c->has_column_program = seen[R0H_SEC_WITGEN] && any_accum;
  R0H_REQUIRE(seen[R0H_SEC_WITGEN] == any_accum && (int)seen[R0H_SEC_ACCUM] + (int)seen[R0H_SEC_ACCUM_FP] + (int)seen[R0H_SEC_LOGUP] <= 1,
              "circuit blob: WITGEN comes with exactly one of ACCUM / ACCUM_FP / LOGUP");
  R0H_REQUIRE(c->n_late <= c->n_global, "circuit blob: more late public inputs than public inputs");
  for (const CodeCol& cc : c->code_cols)
    if (cc.kind == 6) R0H_REQUIRE(c->period && (uint64_t)cc.param * c->period + c->period <= c->periodic.size(), "circuit blob: a periodic CODE column names no column of the PERIODIC section");
  if (c->has_sponge)
    R0H_REQUIRE(c->has_column_program && c->period == R0H_SPONGE_PERIOD && (uint64_t)c->sponge_code + R0H_SPONGE_CODE_COLUMNS <= c->group_size[R0H_GROUP_CODE] &&
                    (uint64_t)c->sponge_data + R0H_SPONGE_DATA_COLUMNS <= c->group_size[R0H_GROUP_DATA] && (uint64_t)c->sponge_global + 8 <= c->n_global && c->global_cols.size() == c->n_global,
                "circuit blob: SPONGE names columns or public inputs outside the circuit");
  if (c->has_column_program) {
    R0H_REQUIRE(c->code_cols.size() == c->group_size[R0H_GROUP_CODE] && c->data_cols.size() == c->group_size[R0H_GROUP_DATA],
                "circuit blob: group sizes disagree with the column programs");
    if (seen[R0H_SEC_ACCUM])
      R0H_REQUIRE(4 * c->acc_cols.size() == c->group_size[R0H_GROUP_ACCUM] && c->n_mix == 8 * c->acc_cols.size(), "circuit blob: group sizes disagree with the accumulators");
    else if (seen[R0H_SEC_LOGUP]) {
      R0H_REQUIRE(4 * c->logup.accs.size() == c->group_size[R0H_GROUP_ACCUM] && c->logup.n_chain >= 1, "circuit blob: group sizes disagree with the log-derivative accumulators");
      auto form_ok = [&](const Lf& lf) {
        for (const LfTerm& t : lf.terms) {
          if (t.global > c->n_global) return false;
          if (t.col) {
            const uint32_t ref = t.col - 1, g = ref >> 28, col = ref & 0xfffffu;
            if ((g != R0H_GROUP_CODE && g != R0H_GROUP_DATA) || col >= c->group_size[g]) return false;
          }
        }
        return true;
      };
      for (const LogupTable& t : c->logup.tables) R0H_REQUIRE(t.data_col < c->group_size[R0H_GROUP_DATA], "circuit blob: LOGUP multiplicity column out of range");
      for (const LogupAcc& a : c->logup.accs) {
        R0H_REQUIRE(a.final_global == 0xffffffffu || (uint64_t)a.final_global + 4 <= c->n_global, "circuit blob: LOGUP total outside the public inputs");
        for (const LogupFraction& f : a.fr) {
          R0H_REQUIRE(form_ok(f.num), "circuit blob: LOGUP numerator refers outside the circuit");
          for (const LogupPart& q : f.parts)
            R0H_REQUIRE(form_ok(q.lf) && (q.ch_kind == 0 || (q.ch_kind == 1 && 4 * (uint64_t)q.ch_idx + 4 <= c->n_mix) || (q.ch_kind == 2 && (uint64_t)q.ch_idx + 4 <= c->n_global)),
                        "circuit blob: LOGUP part refers outside the circuit");
        }
      }
    } else
      R0H_REQUIRE(4 * c->acc_fp.size() == c->group_size[R0H_GROUP_ACCUM] && c->n_mix == 16, "circuit blob: group sizes disagree with the fingerprint accumulators");
  }
  // taps: sorted, in range, every column owns back 0
  std::vector<std::vector<bool>> has0(3);
  for (int g = 0; g < 3; g++) has0[g].assign(c->group_size[g], false);
  for (size_t t = 0; t < c->taps.size(); t++) {
    const Tap& tp = c->taps[t];
    R0H_REQUIRE(tp.group < 3 && tp.offset < c->group_size[tp.group] && tp.back < 64, "circuit blob: tap %zu out of range", t);
    if (t) {
      const Tap& pv = c->taps[t - 1];
      bool ordered = pv.group < tp.group || (pv.group == tp.group && (pv.offset < tp.offset || (pv.offset == tp.offset && pv.back < tp.back)));
      R0H_REQUIRE(ordered, "circuit blob: taps not strictly sorted at %zu", t);
    }
    if (tp.back == 0) has0[tp.group][tp.offset] = true;
  }
  for (int g = 0; g < 3; g++)
    for (uint32_t k = 0; k < c->group_size[g]; k++) R0H_REQUIRE(has0[g][k], "circuit blob: group %d column %u has no back-0 tap", g, k);
  // registers and combos
  c->combo_begin.assign(1, 0);
  for (int g = 0; g < 4; g++) c->group_tap_begin[g] = (uint32_t)c->taps.size();
  for (uint32_t t = 0; t < c->taps.size();) {
    uint32_t e = t;
    while (e < c->taps.size() && c->taps[e].group == c->taps[t].group && c->taps[e].offset == c->taps[t].offset) e++;
    uint32_t size = e - t, n_combos = (uint32_t)c->combo_begin.size() - 1, combo = n_combos;
    for (uint32_t k = 0; k < n_combos && combo == n_combos; k++) {
      uint32_t b = c->combo_begin[k];
      if (c->combo_begin[k + 1] - b != size) continue;
      bool same = true;
      for (uint32_t i = 0; i < size; i++) same = same && c->combo_backs[b + i] == c->taps[t + i].back;
      if (same) combo = k;
    }
    if (combo == n_combos) {
      for (uint32_t i = 0; i < size; i++) c->combo_backs.push_back(c->taps[t + i].back);
      c->combo_begin.push_back((uint32_t)c->combo_backs.size());
    }
    c->regs.push_back(Reg{c->taps[t].group, c->taps[t].offset, t, size, combo});
    t = e;
  }
  for (uint32_t t = (uint32_t)c->taps.size(); t-- > 0;) c->group_tap_begin[c->taps[t].group] = t;
  for (int g = 2; g >= 0; g--)
    if (c->group_tap_begin[g] == c->taps.size()) c->group_tap_begin[g] = c->group_tap_begin[g + 1];
  // variable numbering + operand validation
  for (uint32_t i = 0; i < c->steps.size(); i++) {
    const Step& s = c->steps[i];
    uint32_t nf = (uint32_t)c->fp_step.size(), nm = (uint32_t)c->mix_step.size();
    switch (s.op) {
      case R0H_OP_CONST: R0H_REQUIRE(s.a < P, "poly step %u: constant not canonical", i); c->fp_step.push_back(i); break;
      case R0H_OP_GET: R0H_REQUIRE(s.a < c->taps.size(), "poly step %u: tap out of range", i); c->fp_step.push_back(i); break;
      case R0H_OP_GET_GLOBAL:
        R0H_REQUIRE(s.a < 2 && s.b < (s.a == 0 ? c->n_global : c->n_mix), "poly step %u: global out of range", i);
        c->fp_step.push_back(i);
        break;
      case R0H_OP_ADD: case R0H_OP_SUB: case R0H_OP_MUL:
        R0H_REQUIRE(s.a < nf && s.b < nf, "poly step %u: operand not yet defined", i);
        c->fp_step.push_back(i);
        break;
      case R0H_OP_TRUE: c->mix_step.push_back(i); break;
      case R0H_OP_AND_EQZ: R0H_REQUIRE(s.a < nm && s.b < nf, "poly step %u: operand not yet defined", i); c->mix_step.push_back(i); break;
      case R0H_OP_AND_COND: R0H_REQUIRE(s.a < nm && s.b < nf && s.c < nm, "poly step %u: operand not yet defined", i); c->mix_step.push_back(i); break;
      default: return make_error("poly step %u: unknown opcode %u", i, s.op);
    }
  }
  R0H_REQUIRE(c->ret < c->mix_step.size(), "circuit blob: ret is not a mix variable");
  for (uint32_t k = 0; k < c->n_global && c->has_column_program; k++)
    R0H_REQUIRE(c->global_cols[k] < c->data_cols.size(), "circuit blob: global column out of range");
  for (size_t k = 0; k < c->data_cols.size(); k++) {
    const DataCol& d = c->data_cols[k];
    R0H_REQUIRE(d.kind <= 2, "witgen: data column %zu has unknown kind", k);
    if (d.kind == 0) continue;
    const uint32_t refs[4] = {d.a, d.b, d.kind == 2 ? d.c : d.a, d.e};
    for (uint32_t r : refs) {
      uint32_t g = r >> 28, col = r & 0xfffffu;
      R0H_REQUIRE((g == R0H_GROUP_CODE && col < c->code_cols.size()) || (g == R0H_GROUP_DATA && col < k), "witgen: data column %zu has a forward or foreign reference", k);
    }
  }
  for (const AccCol& a : c->acc_cols) R0H_REQUIRE(a.a < c->data_cols.size() && a.b < c->data_cols.size(), "accum: column out of range");
  for (const AccFp& a : c->acc_fp) {
    R0H_REQUIRE(a.n_f >= 1 && a.n_f <= 3, "accum: a fingerprint accumulator multiplies 1..3 tuples");
    for (uint32_t f = 0; f < 3; f++)
      for (uint32_t q = 0; q < 4; q++) R0H_REQUIRE(a.col[f][q] < c->data_cols.size(), "accum: column out of range");
  }
  c->blob.assign(w, w + n_words);
  return nullptr;
}

// Flatten the MixState chain that ends in `m` into (power, value, gates) terms; returns the number of powers consumed.
static uint32_t flatten(const r0h_circuit* c, uint32_t m, uint32_t base_pow, std::vector<uint32_t>& gates, std::vector<Term>& out) {
  std::vector<uint32_t> chain;
  for (uint32_t cur = m;;) {
    const Step& s = c->steps[c->mix_step[cur]];
    if (s.op == R0H_OP_TRUE) break;
    chain.push_back(cur);
    cur = s.a;
  }
  uint32_t pow = base_pow;
  for (size_t k = chain.size(); k-- > 0;) {
    const Step& s = c->steps[c->mix_step[chain[k]]];
    if (s.op == R0H_OP_AND_EQZ) {
      out.push_back(Term{pow, s.b, gates});
      pow += 1;
    } else {
      gates.push_back(s.b);
      pow += flatten(c, s.c, pow, gates, out);
      gates.pop_back();
    }
  }
  return pow - base_pow;
}

static uint32_t tunable_budget() {
  const char* v = getenv("R0H_EC_BUDGET");
  // expression nodes per kernel; measured on the bench circuit (tools/tune_evalcheck.py): 3,000 -> 11.0 ms, 5,000 -> 10.4, 8,000 -> 10.2,
  // 12,000 -> 9.9, 16,000 -> 9.7 (three kernels, 166 VGPRs), 24,000 -> 11.7 (197 VGPRs: too few waves per SIMD)
  return v && *v ? (uint32_t)strtoul(v, nullptr, 10) : 16000u;
}

static void make_plan(r0h_circuit* c) {
  Plan& pl = c->plan;
  std::vector<uint32_t> gates;
  pl.n_pow = flatten(c, c->ret, 0, gates, pl.terms);
  // cut into kernels of bounded arithmetic: cost of a term = its not-yet-emitted expression nodes + 8
  const uint32_t budget = tunable_budget();
  std::vector<uint32_t> stamp(c->fp_step.size(), UINT32_MAX);
  uint32_t kernel = 0, cost = 0;
  pl.cut.assign(1, 0);
  std::vector<uint32_t> stack;
  for (uint32_t t = 0; t < pl.terms.size(); t++) {
    uint32_t add = 8;
    stack.assign(1, pl.terms[t].v);
    for (uint32_t g : pl.terms[t].conds) stack.push_back(g);
    while (!stack.empty()) {
      uint32_t v = stack.back();
      stack.pop_back();
      if (stamp[v] == kernel) continue;
      stamp[v] = kernel;
      add++;
      const Step& s = c->steps[c->fp_step[v]];
      if (s.op == R0H_OP_ADD || s.op == R0H_OP_SUB || s.op == R0H_OP_MUL) { stack.push_back(s.a); stack.push_back(s.b); }
    }
    if (cost && cost + add > budget) {
      pl.cut.push_back(t);
      kernel++;
      cost = 0;
      t--;  // re-cost this term inside the new kernel
      continue;
    }
    cost += add;
  }
  pl.cut.push_back((uint32_t)pl.terms.size());
}

static const char* PRELUDE = R"SRC(// GENERATED by r0h_circuit_emit_hip -- eval_check for one circuit blob (gfx950).
#if !defined(__HIPCC_RTC__)
#include <hip/hip_runtime.h>
#endif
typedef unsigned int u32;
typedef unsigned long long u64;
#define FP_P 2013265921u
// corrections by p go through the carry flag (v_sub_co_u32 + v_cndmask_b32, full rate) rather than v_min_u32 (half rate on gfx950)
__device__ __forceinline__ u32 fred(u32 x) {
  u32 r;
  asm("v_subrev_co_u32 %0, vcc, 0x78000001, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "=&v"(r) : "v"(x) : "vcc");
  return r;
}
__device__ __forceinline__ u32 fadd(u32 a, u32 b) { return fred(a + b); }
__device__ __forceinline__ u32 fsub(u32 a, u32 b) {
  u32 d, e;
  asm("v_sub_co_u32 %0, vcc, %2, %3\n\tv_add_u32 %1, 0x78000001, %0\n\tv_cndmask_b32 %0, %0, %1, vcc" : "=&v"(d), "=&v"(e) : "v"(a), "v"(b) : "vcc");
  return d;
}
__device__ __forceinline__ u32 fmul(u32 a, u32 b) {
  u64 t = (u64)a * b;
  u32 m = (u32)t * 0x77ffffffu;
  u64 u = t + (u64)m * FP_P;
  return fred((u32)(u >> 32));
}
// Montgomery reduction of a sum of up to four products of reduced words (T < 4 p^2): hi(T) - hi(m*p), m = lo(T) * p^-1;
// hi(T) < 1.875 p, so one conditional subtraction after the sign fix
__device__ __forceinline__ u32 fred64(u64 T) {
  u32 m = (u32)T * 0x88000001u;
  u32 q = __umulhi(m, FP_P);
  u32 h = (u32)(T >> 32);
  return fred(fsub(h, q));
}
// Running 64-bit sums of products of reduced words: four products fit as they are (4 p^2 < 2^64); from then on the high
// word is brought below p before every second product (total < p 2^32 + 2 p^2 < 2^64, high word < 2p).  The empty asm keeps
// the re-packed pair opaque, so the following multiply-adds stay single v_mad_u64_u32 on that pair.
__device__ __forceinline__ u64 dfix(u64 T) {
  T = ((u64)fred((u32)(T >> 32)) << 32) | (u32)T;
  asm("" : "+v"(T));
  return T;
}
__device__ __forceinline__ u32 dfinish(u64 T) { return fred64(dfix(T)); }
#define TAP(g, col, back) g[(size_t)(col) * domain + ((i - 4u * (back)) & mask)]
)SRC";

// Additions whose every use is an operand of a product need no correction: a + b < 2p is a valid Montgomery operand as long as
// the other operand is reduced (a b + 2^32 p < 2^64 still holds, the product comes out below 2p and is corrected as usual).
static std::vector<bool> lazy_additions(const r0h_circuit* c) {
  const size_t nf = c->fp_step.size();
  std::vector<bool> lazy(nf, false), needs_reduced(nf, false);
  for (size_t v = 0; v < nf; v++) lazy[v] = c->steps[c->fp_step[v]].op == R0H_OP_ADD;
  for (const Step& s : c->steps) {
    if (s.op == R0H_OP_ADD || s.op == R0H_OP_SUB) needs_reduced[s.a] = needs_reduced[s.b] = true;
    if (s.op == R0H_OP_AND_EQZ || s.op == R0H_OP_AND_COND) needs_reduced[s.b] = true;  // constraint values and gates
  }
  for (const Term& t : c->plan.terms) {
    needs_reduced[t.v] = true;
    for (uint32_t g : t.conds) needs_reduced[g] = true;
  }
  for (size_t v = 0; v < nf; v++)
    if (needs_reduced[v]) lazy[v] = false;
  for (const Step& s : c->steps)  // at most one uncorrected operand per product
    if (s.op == R0H_OP_MUL) {
      if (s.a == s.b) lazy[s.a] = false;
      else if (lazy[s.a] && lazy[s.b]) lazy[s.b] = false;
    }
  return lazy;
}

// Sums of products reduced once.  An ADD tree whose inner nodes have no other reader and whose leaves include two or more products
// that have no other reader either -- a0 b0 + a1 b1 (+ ...) (+ other terms) -- is emitted as ONE Montgomery reduction of the 64-bit sum
// of those products (fred64: up to four products of reduced words, T < 4 p^2; a product with an uncorrected operand counts double),
// the other leaves added afterwards: per product fused, a v_mul_lo_u32, a v_mad_u64_u32 and a conditional subtraction less.  Same
// canonical word as the step-by-step form (both are the sum mod p, fully reduced).  In the bench circuit: the 2,612 padded
// constraints t0 t1 + t2 t3 + ..
// MEASURED AND LEFT OFF (round 3, profiles/r03/eval_check_fusion.md): fewer multiplier instructions, but the fused form keeps four
// factors and a 64-bit sum live per constraint -- 242 / 230 / 190 VGPRs instead of 165 / 166 / 150 for the three kernels of `bench`,
// two waves per SIMD instead of three -- and eval_check runs 8 % SLOWER (11.05 vs 10.26 ms); cut into eight kernels that fit 162
// VGPRs it is still 5 % slower.  R0H_EC_FUSION=1 turns it on for experiments (tools/tune_evalcheck.py).
struct Fuse { std::vector<uint32_t> muls, rest; };
static std::vector<Fuse> plan_fusion(const r0h_circuit* c, const std::vector<bool>& lazy) {
  const size_t nf = c->fp_step.size();
  std::vector<Fuse> fuse(nf);
  if (!getenv("R0H_EC_FUSION")) return fuse;
  std::vector<uint32_t> uses(nf, 0);
  std::vector<bool> read_by_add_only(nf, true);  // every reader is an ADD (only then can the node dissolve into its reader's tree)
  for (const Step& s : c->steps) {
    if (s.op == R0H_OP_ADD || s.op == R0H_OP_SUB || s.op == R0H_OP_MUL) {
      uses[s.a]++; uses[s.b]++;
      if (s.op != R0H_OP_ADD) read_by_add_only[s.a] = read_by_add_only[s.b] = false;
    }
    if (s.op == R0H_OP_AND_EQZ || s.op == R0H_OP_AND_COND) { uses[s.b]++; read_by_add_only[s.b] = false; }
  }
  auto op_of = [&](uint32_t v) { return c->steps[c->fp_step[v]].op; };
  auto inner = [&](uint32_t v) { return op_of(v) == R0H_OP_ADD && uses[v] == 1 && read_by_add_only[v]; };  // dissolves into its reader
  for (uint32_t v = 0; v < nf; v++) {
    if (op_of(v) != R0H_OP_ADD || inner(v)) continue;  // only the root of a tree is emitted
    Fuse f;
    uint32_t weight = 0;
    std::vector<uint32_t> stack{v};
    while (!stack.empty()) {
      const uint32_t x = stack.back();
      stack.pop_back();
      const Step& s = c->steps[c->fp_step[x]];
      for (uint32_t y : {s.a, s.b}) {
        if (inner(y)) { stack.push_back(y); continue; }
        const Step& m = c->steps[c->fp_step[y]];
        const uint32_t w = op_of(y) == R0H_OP_MUL ? ((lazy[m.a] || lazy[m.b]) ? 2u : 1u) : 0u;
        if (w && uses[y] == 1 && read_by_add_only[y] && weight + w <= 3) { f.muls.push_back(y); weight += w; }
        else f.rest.push_back(y);
      }
    }
    if (f.muls.size() >= 2) fuse[v] = f;
  }
  return fuse;
}

static void emit_var(const r0h_circuit* c, uint32_t root, std::vector<bool>& done, const std::vector<bool>& lazy, const std::vector<Fuse>& fuse,
                     std::ostringstream& os) {
  // iterative post-order emission of the expression DAG below `root`
  std::vector<std::pair<uint32_t, int>> stack;
  stack.push_back({root, 0});
  while (!stack.empty()) {
    auto [v, state] = stack.back();
    if (done[v]) { stack.pop_back(); continue; }
    const Step& s = c->steps[c->fp_step[v]];
    const Fuse& f = fuse[v];
    bool binary = s.op == R0H_OP_ADD || s.op == R0H_OP_SUB || s.op == R0H_OP_MUL;
    if (binary && state == 0) {
      stack.back().second = 1;
      if (!f.muls.empty()) {  // the factors of the fused products and the other leaves, not the tree's own nodes
        for (uint32_t y : f.rest) if (!done[y]) stack.push_back({y, 0});
        for (uint32_t m : f.muls) {
          const Step& ms = c->steps[c->fp_step[m]];
          if (!done[ms.b]) stack.push_back({ms.b, 0});
          if (!done[ms.a]) stack.push_back({ms.a, 0});
        }
        continue;
      }
      if (!done[s.b]) stack.push_back({s.b, 0});
      if (!done[s.a]) stack.push_back({s.a, 0});
      continue;
    }
    stack.pop_back();
    done[v] = true;
    os << "  const u32 v" << v << " = ";
    if (!f.muls.empty()) {
      std::ostringstream sum;
      sum << "fred64(";
      for (size_t k = 0; k < f.muls.size(); k++) {
        const Step& ms = c->steps[c->fp_step[f.muls[k]]];
        sum << (k ? " + " : "") << "(u64)v" << ms.a << " * v" << ms.b;
      }
      sum << ")";
      std::string expr = sum.str();
      for (uint32_t y : f.rest) expr = "fadd(" + expr + ", v" + std::to_string(y) + ")";
      os << expr << ";\n";
      continue;
    }
    switch (s.op) {
      case R0H_OP_CONST: os << enc(s.a) << "u"; break;
      case R0H_OP_GET: {
        const Tap& t = c->taps[s.a];
        os << "TAP(g" << t.group << ", " << t.offset << "u, " << t.back << "u)";
        break;
      }
      case R0H_OP_GET_GLOBAL: os << (s.a == 0 ? "glob[" : "mix[") << s.b << "]"; break;
      case R0H_OP_ADD:
        if (lazy[v]) os << "v" << s.a << " + v" << s.b;  // < 2p: only ever multiplied
        else os << "fadd(v" << s.a << ", v" << s.b << ")";
        break;
      case R0H_OP_SUB: os << "fsub(v" << s.a << ", v" << s.b << ")"; break;
      case R0H_OP_MUL: os << "fmul(v" << s.a << ", v" << s.b << ")"; break;
      default: break;
    }
    os << ";\n";
  }
}

// Code-generation tunables (environment overrides exist for experiments; defaults are the measured best):
//   R0H_EC_BUDGET  expression nodes per kernel          R0H_EC_SCOPE  terms per register scope (0 = one scope)
//   R0H_EC_WAVES   __launch_bounds__ waves/SIMD hint (0 = none)
//   R0H_EC_FUSION  sums of products share one reduction (plan_fusion; measured slower, off)
static uint32_t tunable(const char* name, uint32_t dflt) {
  const char* v = getenv(name);
  return v && *v ? (uint32_t)strtoul(v, nullptr, 10) : dflt;
}

// tot += poly_mix^pow * w as four running 64-bit sums (one per extension component), see dfix in the prelude
static void emit_accumulate(const Plan& pl, uint32_t term, uint32_t position, std::ostringstream& os) {
  if (position >= 4 && position % 2 == 0) os << "  T0 = dfix(T0); T1 = dfix(T1); T2 = dfix(T2); T3 = dfix(T3);\n";
  for (int q = 0; q < 4; q++)
    os << "  T" << q << (position ? " += " : " = ") << "(u64)mixpow[" << 4 * (size_t)pl.terms[term].pow + q << "] * w" << term << ";\n";
}

static std::string emit_source(const r0h_circuit* c) {
  const Plan& pl = c->plan;
  const std::vector<bool> lazy = lazy_additions(c);
  const std::vector<Fuse> fuse = plan_fusion(c, lazy);
  const uint32_t scope_terms = tunable("R0H_EC_SCOPE", 0), waves = tunable("R0H_EC_WAVES", 0);
  std::ostringstream os;
  os << PRELUDE;
  os << "// terms: " << pl.terms.size() << ", powers of poly_mix: " << pl.n_pow << ", kernels: " << pl.cut.size() - 1 << "\n";
  for (size_t k = 0; k + 1 < pl.cut.size(); k++) {
    os << "extern \"C\" __global__ __launch_bounds__(256";
    if (waves) os << ", " << waves;
    os << ") void eval_check_" << k
       << "(u32* __restrict__ check, const u32* __restrict__ g0, const u32* __restrict__ g1, const u32* __restrict__ g2,\n"
          "    const u32* __restrict__ glob, const u32* __restrict__ mix, const u32* __restrict__ mixpow,\n"
          "    const u32* __restrict__ inv_van, u32 po2, u32 accumulate) {\n"
          "  const u32 domain = 4u << po2, mask = domain - 1u;\n"
          "  const u32 i = blockIdx.x * 256u + threadIdx.x;\n"
          "  u64 T0 = 0, T1 = 0, T2 = 0, T3 = 0;\n";
    // Terms are emitted in register scopes: every scope re-loads the taps and re-derives the sub-expressions it needs, and
    // a scheduling barrier keeps the compiler from hoisting the next scope's loads, so the live set is bounded by the
    // scope, not by the circuit.
    std::vector<bool> done(c->fp_step.size(), false);
    uint32_t in_scope = 0;
    os << "  {\n";
    for (uint32_t t = pl.cut[k]; t < pl.cut[k + 1]; t++) {
      if (scope_terms && in_scope == scope_terms) {
        os << "  }\n  __builtin_amdgcn_sched_barrier(0);\n  {\n";
        std::fill(done.begin(), done.end(), false);
        in_scope = 0;
      }
      const Term& tm = pl.terms[t];
      emit_var(c, tm.v, done, lazy, fuse, os);
      for (uint32_t g : tm.conds) emit_var(c, g, done, lazy, fuse, os);
      // value of the term: the constraint times its enclosing gates
      std::ostringstream val;
      for (size_t g = 0; g < tm.conds.size(); g++) val << "fmul(v" << tm.conds[g] << ", ";
      val << "v" << tm.v;
      for (size_t g = 0; g < tm.conds.size(); g++) val << ")";
      os << "  const u32 w" << t << " = " << val.str() << ";\n";
      emit_accumulate(pl, t, t - pl.cut[k], os);
      in_scope++;
    }
    os << "  }\n";
    os << "  u32 t0 = dfinish(T0), t1 = dfinish(T1), t2 = dfinish(T2), t3 = dfinish(T3);\n";
    os << "  const u32 iv = inv_van[i & 3u];\n"
          "  t0 = fmul(t0, iv); t1 = fmul(t1, iv); t2 = fmul(t2, iv); t3 = fmul(t3, iv);\n"
          "  if (accumulate) {\n"
          "    t0 = fadd(t0, check[i]); t1 = fadd(t1, check[(size_t)domain + i]);\n"
          "    t2 = fadd(t2, check[2 * (size_t)domain + i]); t3 = fadd(t3, check[3 * (size_t)domain + i]);\n"
          "  }\n"
          "  check[i] = t0; check[(size_t)domain + i] = t1; check[2 * (size_t)domain + i] = t2; check[3 * (size_t)domain + i] = t3;\n"
          "}\n\n";
  }
  return os.str();
}

static const char* compile_in_process(const std::string& src, std::vector<char>& code) {
  hiprtcProgram prog;
  hiprtcResult r = hiprtcCreateProgram(&prog, src.c_str(), "eval_check.hip", 0, nullptr, nullptr);
  R0H_REQUIRE(r == HIPRTC_SUCCESS, "hiprtcCreateProgram: %s", hiprtcGetErrorString(r));
  const char* opts[] = {"--offload-arch=gfx950", "-O3"};
  r = hiprtcCompileProgram(prog, 2, opts);
  if (r != HIPRTC_SUCCESS) {
    size_t n = 0;
    hiprtcGetProgramLogSize(prog, &n);
    std::string log(n, 0);
    if (n) hiprtcGetProgramLog(prog, &log[0]);
    hiprtcDestroyProgram(&prog);
    return make_error("hiprtcCompileProgram: %s\n%.2000s", hiprtcGetErrorString(r), log.c_str());
  }
  size_t n = 0;
  hiprtcGetCodeSize(prog, &n);
  code.resize(n);
  hiprtcGetCode(prog, code.data());
  hiprtcDestroyProgram(&prog);
  return nullptr;
}

// ------------------------------------------------------------------ synthetic witness + accumulation kernels
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__device__ __forceinline__ uint32_t synth_word(uint64_t seed_mixed, uint32_t stream, uint32_t row) {
  uint64_t h = splitmix64(seed_mixed ^ (((uint64_t)stream << 32) | row));
  return (uint32_t)(((h >> 32) * (uint64_t)P) >> 32);
}
static uint64_t splitmix64_host(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

__global__ void witgen_fixed_kernel(uint32_t* __restrict__ buf, const uint32_t* __restrict__ kinds /* (kind, stream) per col */,
                                    uint32_t po2, uint64_t seed_mixed, const uint32_t* __restrict__ periodic /* Montgomery */, uint32_t period) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x, n = 1u << po2, col = blockIdx.y;
  const uint32_t kind = kinds[2 * col], stream = kinds[2 * col + 1];
  uint32_t v;
  if (kind == 0) v = r == 0 ? ONE : 0u;
  else if (kind == 1) v = r == n - 1 ? ONE : 0u;
  else if (kind == 2) v = enc(r);
  else if (kind == 3) v = synth_word(seed_mixed, stream, r);
  else if (kind == 4) v = r < 65536u ? enc(r) : 0u;                                                     // the 16-bit range table
  else if (kind == 5) v = enc(R0H_TAG_AND + (r < 65536u ? r + 65536u * ((r & 255u) & (r >> 8)) : 0u));  // the byte-AND table
  else if (kind == 6) v = r < (n / period) * period ? periodic[stream * period + r % period] : 0u;      // a periodic schedule (stream = its column)
  else return;  // derived column: filled later
  buf[((size_t)col << po2) + r] = v;
}
__global__ void witgen_derived_kernel(uint32_t* __restrict__ dst, const uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                                      const uint32_t* __restrict__ c, const uint32_t* __restrict__ e, uint32_t ba, uint32_t bb,
                                      uint32_t bc, uint32_t be, uint32_t kind, uint32_t po2) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x, mask = (1u << po2) - 1;
  uint32_t prod = mul(a[(r - ba) & mask], b[(r - bb) & mask]);
  if (kind == 2) prod = mul(prod, c[(r - bc) & mask]);
  dst[r] = add(prod, e[(r - be) & mask]);
}
__global__ void accum_term_kernel(uint32_t* __restrict__ term, const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, Fp4 m0, Fp4 m1) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  Fp4 t = m0 + scale(m1, b[r]);
  t.e[0] = add(t.e[0], a[r]);
  *(uint4*)(term + 4 * (size_t)r) = make_uint4(t.e[0], t.e[1], t.e[2], t.e[3]);
}
// term[r] = prod_{f < n_f} (alpha - addr_f[r] - b1 lo_f[r] - b2 hi_f[r] - b3 t_f[r]): one tuple fingerprint per access (R0H_SEC_ACCUM_FP)
struct FpCols { const uint32_t* col[3][4]; uint32_t n_f; };
__global__ void accum_fp_term_kernel(uint32_t* __restrict__ term, FpCols cols, Fp4 alpha, Fp4 b1, Fp4 b2, Fp4 b3) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  Fp4 prod = fp4_one();
  for (uint32_t f = 0; f < cols.n_f; f++) {
    Fp4 t = alpha - scale(b1, cols.col[f][1][r]) - scale(b2, cols.col[f][2][r]) - scale(b3, cols.col[f][3][r]);
    t.e[0] = sub(t.e[0], cols.col[f][0][r]);
    prod = f ? prod * t : t;
  }
  *(uint4*)(term + 4 * (size_t)r) = make_uint4(prod.e[0], prod.e[1], prod.e[2], prod.e[3]);
}
__global__ void accum_unpack_kernel(uint32_t* __restrict__ cols, const uint32_t* __restrict__ term, uint32_t po2) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  uint4 v = *(const uint4*)(term + 4 * (size_t)r);
  cols[r] = v.x; cols[((size_t)1 << po2) + r] = v.y; cols[((size_t)2 << po2) + r] = v.z; cols[((size_t)3 << po2) + r] = v.w;
}

}  // namespace r0h

using namespace r0h;

extern "C" {

const char* r0h_circuit_emit_hip(const uint32_t* blob, size_t n_words, char** source_out) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(blob && source_out, "r0h_circuit_emit_hip: NULL argument");
  r0h_circuit c;
  R0H_TRY(parse_blob(&c, blob, n_words));
  make_plan(&c);
  std::string src = emit_source(&c);
  char* out = (char*)malloc(src.size() + 1);
  R0H_REQUIRE(out, "r0h_circuit_emit_hip: out of memory");
  memcpy(out, src.c_str(), src.size() + 1);
  *source_out = out;
  return nullptr;
  R0H_GUARD_END
}

const char* r0h_circuit_load(r0h_ctx* ctx, const uint32_t* blob, size_t n_words, const char* code_object_path, r0h_circuit** out) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(ctx && blob && out, "r0h_circuit_load: NULL argument");
  r0h_circuit* c = new r0h_circuit();
  c->ctx = ctx;
  const char* err = parse_blob(c, blob, n_words);
  if (err) { delete c; return err; }
  make_plan(c);
  R0H_TRY_HIP(hipSetDevice(ctx->device));
  std::vector<char> code;
  if (code_object_path) {
    FILE* f = fopen(code_object_path, "rb");
    if (!f) { delete c; return make_error("r0h_circuit_load: cannot open code object %s", code_object_path); }
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    code.resize(sz > 0 ? (size_t)sz : 0);
    size_t got = fread(code.data(), 1, code.size(), f);
    fclose(f);
    if (got != code.size() || code.empty()) { delete c; return make_error("r0h_circuit_load: short read of %s", code_object_path); }
  } else {
    err = compile_in_process(emit_source(c), code);
    if (err) { delete c; return err; }
  }
  hipError_t e = hipModuleLoadData(&c->module, code.data());
  if (e != hipSuccess) { delete c; return make_error("r0h_circuit_load: hipModuleLoadData: %s", hipGetErrorString(e)); }
  for (size_t k = 0;; k++) {  // the code object decides into how many kernels the program was cut
    char name[64];
    snprintf(name, sizeof name, "eval_check_%zu", k);
    hipFunction_t fn;
    if (hipModuleGetFunction(&fn, c->module, name) != hipSuccess) break;
    c->kernels.push_back(fn);
  }
  (void)hipGetLastError();
  if (c->kernels.empty() && !c->plan.terms.empty()) {
    hipModuleUnload(c->module);
    delete c;
    return make_error("r0h_circuit_load: the code object has no eval_check_0 (built from another source?)");
  }
  ctx_retain(ctx);
  *out = c;
  return nullptr;
  R0H_GUARD_END
}

const char* r0h_circuit_free(r0h_circuit* c) {
  if (!c) return nullptr;
  r0h_ctx* ctx = c->ctx;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (c->module) (void)hipModuleUnload(c->module);
  delete c;
  ctx_release(ctx);
  return nullptr;
}

uint32_t r0h_circuit_group_size(const r0h_circuit* c, uint32_t group) { return c && group < 3 ? c->group_size[group] : 0; }
uint32_t r0h_circuit_n_global(const r0h_circuit* c) { return c ? c->n_global : 0; }
uint32_t r0h_circuit_n_mix(const r0h_circuit* c) { return c ? c->n_mix : 0; }
uint32_t r0h_circuit_n_taps(const r0h_circuit* c) { return c ? (uint32_t)c->taps.size() : 0; }

}  // extern "C"
namespace r0h {
const char* sponge_plant(r0h_ctx* ctx, const r0h_circuit* c, uint32_t po2, const uint32_t* words, size_t n_words, r0h_buf* data) {
  R0H_REQUIRE(ctx && c && data && c->has_sponge && (words || !n_words), "sponge_plant: NULL argument, or a circuit without the sponge component");
  const size_t n = (size_t)1 << po2, n_perm = n_words ? (n_words + P2_RATE - 1) / P2_RATE : 1, rows = n_perm * R0H_SPONGE_PERIOD;
  R0H_REQUIRE(rows < n, "the in-circuit sponge over %zu words takes %zu rows: the recursion trace has 2^%u", n_words, rows, po2);
  R0H_REQUIRE(((size_t)c->group_size[R0H_GROUP_DATA] << po2) * 4 <= data->bytes, "sponge_plant: DATA buffer too small for 2^%u rows", po2);
  for (size_t i = 0; i < n_words; i++) R0H_REQUIRE(words[i] < P, "sponge_plant: word %zu is not a canonical field element", i);
  std::unique_ptr<P2Consts> k(new P2Consts);
  p2_default_host(*k);
  std::vector<uint32_t> cols((size_t)R0H_SPONGE_DATA_COLUMNS * rows);
  size_t used = 0;
  p2_sponge_rows_host(*k, words, n_words, cols.data(), rows, &used);
  uint32_t* first = u32(data) + ((size_t)c->sponge_data << po2);
  R0H_TRY_HIP(hipMemsetAsync(first, 0, (size_t)R0H_SPONGE_DATA_COLUMNS * n * 4, ctx->stream));
  R0H_TRY_HIP(hipMemcpy2DAsync(first, n * 4, cols.data(), rows * 4, rows * 4, R0H_SPONGE_DATA_COLUMNS, hipMemcpyHostToDevice, ctx->stream));
  R0H_TRY_HIP(hipStreamSynchronize(ctx->stream));  // `cols` is pageable and goes out of scope
  return nullptr;
}
}  // namespace r0h
extern "C" {

static const char* witgen_impl(r0h_ctx* ctx, const r0h_circuit* c, uint32_t po2, uint64_t seed, const uint32_t* global_in, r0h_buf* code,
                               r0h_buf* data, uint32_t* global_out) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(ctx && c && code && data, "r0h_witgen: NULL argument");
  R0H_REQUIRE(c->has_column_program, "r0h_witgen: this circuit carries no synthetic column program (WITGEN/ACCUM sections); supply the witness");
  R0H_REQUIRE(po2 >= 4 && po2 <= R0H_MAX_PO2, "r0h_witgen: po2 %u outside [4, %u]", po2, R0H_MAX_PO2);
  const uint32_t n = 1u << po2, threads = n < 256 ? n : 256, nc = (uint32_t)c->code_cols.size(), nd = (uint32_t)c->data_cols.size();
  R0H_REQUIRE(((size_t)nc << po2) * 4 <= code->bytes && ((size_t)nd << po2) * 4 <= data->bytes, "r0h_witgen: buffers too small for 2^%u rows", po2);
  std::vector<uint32_t> kinds(2 * (size_t)(nc + nd));
  for (uint32_t k = 0; k < nc; k++) { kinds[2 * k] = c->code_cols[k].kind; kinds[2 * k + 1] = c->code_cols[k].kind == 6 ? c->code_cols[k].param : (1u << 16) | k; }
  for (uint32_t k = 0; k < nd; k++) { kinds[2 * (nc + k)] = c->data_cols[k].kind == 0 ? 3u : 99u; kinds[2 * (nc + k) + 1] = (2u << 16) | k; }
  const size_t table_at = kinds.size();
  for (uint32_t v : c->periodic) kinds.push_back(enc(v));
  R0H_TRY(ensure_scratch(ctx, kinds.size() * 4));
  R0H_TRY(stage_h2d(ctx, ctx->scratch, kinds.data(), kinds.size() * 4));
  const uint32_t* dk = (const uint32_t*)ctx->scratch;
  hipLaunchKernelGGL(witgen_fixed_kernel, dim3(n / threads, nc), dim3(threads), 0, ctx->stream, u32(code), dk, po2, splitmix64_host(0xC0DEull), dk + table_at, c->period);
  hipLaunchKernelGGL(witgen_fixed_kernel, dim3(n / threads, nd), dim3(threads), 0, ctx->stream, u32(data), dk + 2 * nc, po2, splitmix64_host(seed), dk + table_at, c->period);
  if (c->has_sponge) {
    // the sponge's columns hold the sponge over no words at all, and -- unless the caller names the public inputs -- the inputs its
    // digest is tied to are that digest (a caller who names them plants the rows of what it hashed instead: r0h_lift / r0h_join)
    R0H_TRY(sponge_plant(ctx, c, po2, nullptr, 0, data));
    if (!global_in) {
      std::unique_ptr<P2Consts> k(new P2Consts);
      p2_default_host(*k);
      uint32_t digest[8];
      p2_hash_elems_host(*k, nullptr, 0, digest);
      for (uint32_t j = 0; j < 8; j++) R0H_TRY(stage_h2d(ctx, u32(data) + ((size_t)c->global_cols[c->sponge_global + j] << po2), digest + j, 4));
    }
  }
  if (global_in) {  // caller-chosen public inputs: row 0 of the (free) columns the globals are read from, before anything is derived from them
    for (uint32_t k = 0; k < c->n_global; k++) {
      R0H_REQUIRE(global_in[k] < P, "r0h_witgen_public: global %u is not a canonical field word", k);
      R0H_TRY(stage_h2d(ctx, u32(data) + ((size_t)c->global_cols[k] << po2), global_in + k, 4));
    }
  }
  for (uint32_t k = 0; k < nd; k++) {
    const DataCol& d = c->data_cols[k];
    if (d.kind == 0) continue;
    auto col = [&](uint32_t r) { return ((r >> 28) == R0H_GROUP_CODE ? u32(code) : u32(data)) + ((size_t)(r & 0xfffffu) << po2); };
    auto back = [](uint32_t r) { return (r >> 20) & 0xffu; };
    uint32_t rc = d.kind == 2 ? d.c : d.a;
    hipLaunchKernelGGL(witgen_derived_kernel, dim3(n / threads), dim3(threads), 0, ctx->stream, u32(data) + ((size_t)k << po2), col(d.a), col(d.b),
                       col(rc), col(d.e), back(d.a), back(d.b), back(rc), back(d.e), d.kind, po2);
  }
  hipError_t e = hipGetLastError();
  R0H_REQUIRE(e == hipSuccess, "r0h_witgen: launch failed: %s", hipGetErrorString(e));
  if (global_out) {
    for (uint32_t k = 0; k < c->n_global; k++)
      R0H_TRY_HIP(hipMemcpyAsync(global_out + k, u32(data) + ((size_t)c->global_cols[k] << po2), 4, hipMemcpyDeviceToHost, ctx->stream));
  }
  R0H_TRY_HIP(hipStreamSynchronize(ctx->stream));
  return nullptr;
  R0H_GUARD_END
}

const char* r0h_witgen(r0h_ctx* ctx, const r0h_circuit* c, uint32_t po2, uint64_t seed, r0h_buf* code, r0h_buf* data, uint32_t* global_out) {
  return witgen_impl(ctx, c, po2, seed, nullptr, code, data, global_out);
}

const char* r0h_witgen_public(r0h_ctx* ctx, const r0h_circuit* c, uint32_t po2, uint64_t seed, const uint32_t* global_in, r0h_buf* code, r0h_buf* data) {
  R0H_REQUIRE(global_in || r0h_circuit_n_global(c) == 0, "r0h_witgen_public: global_in is NULL");
  return witgen_impl(ctx, c, po2, seed, global_in, code, data, nullptr);
}

const char* r0h_accum(r0h_ctx* ctx, const r0h_circuit* c, uint32_t po2, const r0h_buf* code, const r0h_buf* data, const uint32_t* mix, r0h_buf* accum) {
  R0H_GUARD_BEGIN
  (void)code;
  R0H_REQUIRE(ctx && c && data && accum && (mix || c->n_mix == 0), "r0h_accum: NULL argument");
  R0H_REQUIRE(c->has_column_program, "r0h_accum: this circuit carries no synthetic column program (WITGEN/ACCUM sections)");
  R0H_REQUIRE(po2 >= 4 && po2 <= R0H_MAX_PO2, "r0h_accum: po2 %u outside [4, %u]", po2, R0H_MAX_PO2);
  const uint32_t n = 1u << po2, threads = n < 256 ? n : 256;
  R0H_REQUIRE(((size_t)c->data_cols.size() << po2) * 4 <= data->bytes && ((size_t)c->group_size[R0H_GROUP_ACCUM] << po2) * 4 <= accum->bytes,
              "r0h_accum: buffers too small for 2^%u rows", po2);
  for (uint32_t i = 0; i < c->n_mix; i++) R0H_REQUIRE(mix[i] < P, "r0h_accum: mix[%u] not canonical", i);
  R0H_REQUIRE(c->logup.accs.empty(), "r0h_accum: this circuit accumulates a log-derivative argument that reads public inputs: use r0h_accum_public");
  r0h_buf* term = nullptr;
  R0H_TRY(buf_alloc_pooled(ctx, (size_t)n * 16, &term));
  for (uint32_t j = 0; j < c->acc_fp.size(); j++) {
    const AccFp& a = c->acc_fp[j];
    FpCols cols;
    cols.n_f = a.n_f;
    for (uint32_t f = 0; f < 3; f++)
      for (uint32_t q = 0; q < 4; q++) cols.col[f][q] = u32(data) + ((size_t)a.col[f][q] << po2);
    auto m = [&](uint32_t k) { return Fp4{{mix[4 * k], mix[4 * k + 1], mix[4 * k + 2], mix[4 * k + 3]}}; };
    hipLaunchKernelGGL(accum_fp_term_kernel, dim3(n / threads), dim3(threads), 0, ctx->stream, u32(term), cols, m(0), m(1), m(2), m(3));
    const char* err = r0h_prefix_products(ctx, term, n);
    if (err) { r0h_buf_free(term); return err; }
    hipLaunchKernelGGL(accum_unpack_kernel, dim3(n / threads), dim3(threads), 0, ctx->stream, u32(accum) + ((size_t)(4 * j) << po2), u32(term), po2);
  }
  for (uint32_t j = 0; j < c->acc_cols.size(); j++) {
    Fp4 m0{{mix[8 * j], mix[8 * j + 1], mix[8 * j + 2], mix[8 * j + 3]}}, m1{{mix[8 * j + 4], mix[8 * j + 5], mix[8 * j + 6], mix[8 * j + 7]}};
    hipLaunchKernelGGL(accum_term_kernel, dim3(n / threads), dim3(threads), 0, ctx->stream, u32(term), u32(data) + ((size_t)c->acc_cols[j].a << po2),
                       u32(data) + ((size_t)c->acc_cols[j].b << po2), m0, m1);
    const char* err = r0h_prefix_products(ctx, term, n);
    if (err) { r0h_buf_free(term); return err; }
    hipLaunchKernelGGL(accum_unpack_kernel, dim3(n / threads), dim3(threads), 0, ctx->stream, u32(accum) + ((size_t)(4 * j) << po2), u32(term), po2);
  }
  hipError_t e = hipGetLastError();
  R0H_TRY(r0h_buf_free(term));
  R0H_REQUIRE(e == hipSuccess, "r0h_accum: launch failed: %s", hipGetErrorString(e));
  return nullptr;
  R0H_GUARD_END
}

const char* r0h_accum_public(r0h_ctx* ctx, const r0h_circuit* c, uint32_t po2, const r0h_buf* code, const r0h_buf* data, const uint32_t* global, const uint32_t* mix, r0h_buf* accum) {
  R0H_REQUIRE(ctx && c, "r0h_accum_public: NULL argument");
  if (c->logup.accs.empty()) return r0h_accum(ctx, c, po2, code, data, mix, accum);
  return logup_accum(ctx, c, po2, code, data, global, mix, accum);
}

const char* r0h_eval_check(r0h_ctx* ctx, const r0h_circuit* c, uint32_t po2, const r0h_buf* eval_accum, const r0h_buf* eval_code,
                           const r0h_buf* eval_data, const uint32_t* global, const uint32_t* mix, const uint32_t poly_mix[4], r0h_buf* check) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(ctx && c && eval_accum && eval_code && eval_data && check && poly_mix, "r0h_eval_check: NULL argument");
  R0H_REQUIRE((global || !c->n_global) && (mix || !c->n_mix), "r0h_eval_check: NULL globals");
  R0H_REQUIRE(po2 >= 6 && po2 <= R0H_MAX_PO2, "r0h_eval_check: po2 %u outside [6, %u]", po2, R0H_MAX_PO2);
  for (int q = 0; q < 4; q++) R0H_REQUIRE(poly_mix[q] < P, "r0h_eval_check: poly_mix words must be canonical (< p)");
  const size_t domain = (size_t)4 << po2;
  const r0h_buf* g[3] = {eval_accum, eval_code, eval_data};
  for (int k = 0; k < 3; k++) R0H_REQUIRE(domain * c->group_size[k] * 4 <= g[k]->bytes, "r0h_eval_check: group %d buffer too small", k);
  R0H_REQUIRE(domain * 16 <= check->bytes, "r0h_eval_check: check buffer too small");
  // parameters: globals, mix, powers of poly_mix, inverse vanishing values
  std::vector<uint32_t> params(c->n_global + c->n_mix + 4 * (size_t)c->plan.n_pow + 4);
  uint32_t* p = params.data();
  for (uint32_t i = 0; i < c->n_global; i++) { R0H_REQUIRE(global[i] < P, "r0h_eval_check: global[%u] not canonical", i); *p++ = global[i]; }
  for (uint32_t i = 0; i < c->n_mix; i++) { R0H_REQUIRE(mix[i] < P, "r0h_eval_check: mix[%u] not canonical", i); *p++ = mix[i]; }
  Fp4 pm{{poly_mix[0], poly_mix[1], poly_mix[2], poly_mix[3]}}, cur = fp4_one();
  for (uint32_t k = 0; k < c->plan.n_pow; k++) { memcpy(p, cur.e, 16); p += 4; cur = cur * pm; }
  {
    uint32_t three_n = fpow(enc(3), (uint64_t)1 << po2), w4 = rou_fwd(2), w = ONE;
    for (int k = 0; k < 4; k++) { *p++ = inv(sub(mul(three_n, w), ONE)); w = mul(w, w4); }
  }
  // the parameter block comes from the calling context's pool (stream-ordered reuse), not from the circuit: one loaded circuit
  // then serves every context of its device at once (r0h_prove_elf's prover lanes share it)
  r0h_buf* pbuf = nullptr;
  R0H_TRY(buf_alloc_pooled(ctx, params.size() * 4, &pbuf));
  struct Free { r0h_buf* b; ~Free() { r0h_buf_free(b); } } pguard{pbuf};
  R0H_TRY(stage_h2d(ctx, pbuf->ptr, params.data(), params.size() * 4));
  uint32_t* d_check = u32(check);
  const uint32_t *g0 = u32(g[0]), *g1 = u32(g[1]), *g2 = u32(g[2]);
  const uint32_t *d_glob = u32(pbuf), *d_mix = d_glob + c->n_global, *d_pow = d_mix + c->n_mix, *d_van = d_pow + 4 * (size_t)c->plan.n_pow;
  double alg = (double)domain * 16;
  for (int k = 0; k < 3; k++) alg += (double)domain * c->group_size[k] * 4;
  KScope ks(ctx, "eval_check", alg);
  for (size_t k = 0; k < c->kernels.size(); k++) {
    uint32_t accumulate = k ? 1u : 0u;
    void* args[] = {&d_check, &g0, &g1, &g2, &d_glob, &d_mix, &d_pow, &d_van, &po2, &accumulate};
    R0H_TRY_HIP(hipModuleLaunchKernel(c->kernels[k], (uint32_t)(domain / 256), 1, 1, 256, 1, 1, 0, ctx->stream, args, nullptr));
  }
  if (c->kernels.empty()) R0H_TRY_HIP(hipMemsetAsync(check->ptr, 0, domain * 16, ctx->stream));
  return nullptr;
  R0H_GUARD_END
}

}  // extern "C"
