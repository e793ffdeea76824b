// The log-derivative accumulation (blob section LOGUP, include/r0hip_circuit.h): the ACCUM group of a circuit whose argument is a sum
// of fractions numerator / (challenge-weighted linear forms of the row) -- lookups into the CODE group's tables, memory tuples,
// session tuples.  Stands where risc0-circuit-rv32im 4.0.4's `step_accum` + the hal's `prefix_products` stand (SURVEY.md 8(a) a10:
// "lookup/permutation argument ... log-derivative"); the fractions themselves come from the blob (tools/trace_circuit.py).
//
// Three device steps, all streams over columns (consecutive lanes own consecutive rows, every column access is a 256-byte line):
//   1. multiplicities: every lookup's value is binned -- per workgroup in LDS for the small (hot) values, with device atomics for the
//      rest; zero, by far the most frequent value (idle slots), is counted by subtraction -- and the counts become the table's
//      multiplicity column in DATA (before DATA is committed);
//   2. terms: one thread per row evaluates the row's fractions accumulator by accumulator, sum_f n_f / d_f = N / D with ONE
//      extension-field inversion per accumulator, and leaves the running sum WITHIN the row in the ACCUM columns;
//   3. the row totals are scanned (r0h_prefix_sums) and added back: the chain runs through the rows.
// The interpreter reads a flat tape of the fractions (uniform across the wave: scalar loads) built per launch on the host, public
// inputs folded into the coefficients.
#include <algorithm>
#include <map>

#include "../../include/r0hip_circuit.h"
#include "circuit.hpp"

namespace r0h {

namespace {
struct Tape {
  std::vector<uint32_t> words;              // per accumulator, per fraction: table, num form, n_parts, (challenge index, form)...; form = n, (coef, column + 1)...
  std::vector<const uint32_t*> cols;        // column base pointers
  std::vector<Fp4> ch;                      // challenges; index 0 is "one"
  std::vector<uint32_t> acc_begin;          // word offset of every accumulator
};

const char* build_tape(const r0h_circuit* c, uint32_t po2, const r0h_buf* code, const r0h_buf* data, const uint32_t* global, const uint32_t* mix, Tape* t, bool lookups_only = false) {
  std::map<uint32_t, uint32_t> col_index;
  std::map<uint64_t, uint32_t> ch_index;
  t->ch.push_back(fp4_one());
  auto form = [&](const Lf& lf) -> const char* {
    t->words.push_back((uint32_t)lf.terms.size());
    for (const LfTerm& term : lf.terms) {
      uint32_t coef = enc(term.coef);
      if (term.global) {
        R0H_REQUIRE(global, "log-derivative accumulation: a form reads a public input and none were given");
        coef = mul(coef, global[term.global - 1]);
      }
      uint32_t col = 0;
      if (term.col) {
        const uint32_t ref = term.col - 1;
        auto it = col_index.find(ref);
        if (it == col_index.end()) {
          const r0h_buf* src = (ref >> 28) == R0H_GROUP_CODE ? code : data;
          R0H_REQUIRE(src, "log-derivative accumulation: a form reads the %s group and none was given", (ref >> 28) == R0H_GROUP_CODE ? "CODE" : "DATA");
          it = col_index.emplace(ref, (uint32_t)t->cols.size()).first;
          t->cols.push_back(u32(src) + ((size_t)(ref & 0xfffffu) << po2));
        }
        col = it->second + 1;
      }
      t->words.push_back(coef);
      t->words.push_back(col);
    }
    return nullptr;
  };
  for (const LogupAcc& a : c->logup.accs) {
    t->acc_begin.push_back((uint32_t)t->words.size());
    for (const LogupFraction& f : a.fr) {
      if (lookups_only && !f.table) continue;
      t->words.push_back(f.table);
      R0H_TRY(form(f.num));
      t->words.push_back((uint32_t)f.parts.size());
      for (const LogupPart& q : f.parts) {
        uint32_t idx = 0;
        if (q.ch_kind) {
          const uint64_t key = (uint64_t)q.ch_kind << 32 | q.ch_idx;
          auto it = ch_index.find(key);
          if (it == ch_index.end()) {
            const uint32_t* src = q.ch_kind == 1 ? mix + 4 * (size_t)q.ch_idx : global + q.ch_idx;
            R0H_REQUIRE(q.ch_kind == 1 ? mix != nullptr : global != nullptr, "log-derivative accumulation: a challenge is read from %s and none were given", q.ch_kind == 1 ? "the mix" : "the public inputs");
            it = ch_index.emplace(key, (uint32_t)t->ch.size()).first;
            t->ch.push_back(Fp4{{src[0], src[1], src[2], src[3]}});
          }
          idx = it->second;
        }
        t->words.push_back(idx);
        R0H_TRY(form(q.lf));
      }
    }
  }
  t->acc_begin.push_back((uint32_t)t->words.size());
  return nullptr;
}

struct DeviceTape {
  r0h_buf* buf = nullptr;
  const uint32_t* words = nullptr;
  const uint32_t* const* cols = nullptr;
  const Fp4* ch = nullptr;
  ~DeviceTape() { if (buf) r0h_buf_free(buf); }
};
const char* upload_tape(r0h_ctx* ctx, const Tape& t, DeviceTape* d) {
  const size_t w_bytes = (t.words.size() * 4 + 15) & ~(size_t)15, c_bytes = (t.cols.size() * sizeof(void*) + 15) & ~(size_t)15, h_bytes = t.ch.size() * 16;
  R0H_TRY(buf_alloc_pooled(ctx, w_bytes + c_bytes + h_bytes, &d->buf));
  char* base = (char*)d->buf->ptr;
  R0H_TRY(stage_h2d(ctx, base, t.words.data(), t.words.size() * 4));
  R0H_TRY(stage_h2d(ctx, base + w_bytes, t.cols.data(), t.cols.size() * sizeof(void*)));
  R0H_TRY(stage_h2d(ctx, base + w_bytes + c_bytes, t.ch.data(), h_bytes));
  d->words = (const uint32_t*)base;
  d->cols = (const uint32_t* const*)(base + w_bytes);
  d->ch = (const Fp4*)(base + w_bytes + c_bytes);
  return nullptr;
}

__device__ __forceinline__ uint32_t eval_form(const uint32_t* __restrict__ tape, uint32_t& at, const uint32_t* const* __restrict__ cols, uint32_t r) {
  const uint32_t n = tape[at++];
  uint32_t acc = 0;
  for (uint32_t k = 0; k < n; k++) {
    const uint32_t coef = tape[at], col = tape[at + 1];
    at += 2;
    acc = add(acc, col ? mul(coef, cols[col - 1][r]) : coef);
  }
  return acc;
}

constexpr uint32_t HOT = 4096;  // values below this are binned in LDS per workgroup

// one thread per row: every lookup's value goes into its table's histogram (zero is not counted: it is the remainder)
__global__ __launch_bounds__(256) void logup_count_kernel(uint32_t* __restrict__ hist /* [n_tables][65536] */, const uint32_t* __restrict__ tape, uint32_t tape_end,
                                                          const uint32_t* const* __restrict__ cols, uint32_t n, uint32_t n_tables) {
  __shared__ uint32_t hot[2][HOT];
  for (uint32_t i = threadIdx.x; i < 2 * HOT; i += 256) (&hot[0][0])[i] = 0;
  __syncthreads();
  const uint32_t r = blockIdx.x * 256u + threadIdx.x;
  if (r < n) {
    uint32_t at = 0;
    while (at < tape_end) {  // fractions in tape order; an accumulator boundary has no marker of its own (four fractions each)
      const uint32_t table = tape[at++];
      const uint32_t num = eval_form(tape, at, cols, r);
      const uint32_t n_parts = tape[at++];
      uint32_t value = 0;
      for (uint32_t q = 0; q < n_parts; q++) {
        at++;  // challenge index
        const uint32_t v = eval_form(tape, at, cols, r);
        if (q == 1) value = v;
      }
      if (!table || table > n_tables || num != ONE) continue;
      uint32_t v = dec(neg(value));
      if (table == R0H_TABLE_AND) {
        v -= R0H_TAG_AND;
        if (v >> 24 || ((v & 255u) & ((v >> 8) & 255u)) != v >> 16) continue;  // not an entry of the table: the sum will not close
        v &= 0xffffu;
      } else if (v >> 16) {
        continue;
      }
      if (!v) continue;
      if (v < HOT) atomicAdd(&hot[table - 1][v], 1u);
      else atomicAdd(&hist[(size_t)(table - 1) * 65536 + v], 1u);
    }
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < 2 * HOT; i += 256) {
    const uint32_t cnt = (&hot[0][0])[i];
    if (cnt) atomicAdd(&hist[(size_t)(i / HOT) * 65536 + (i % HOT)], cnt);
  }
}
// hist -> multiplicity column: entry 0 takes what the other entries leave of `total` lookups
__global__ void logup_mult_kernel(uint32_t* __restrict__ column, const uint32_t* __restrict__ hist, uint32_t total, uint32_t n) {
  __shared__ uint32_t part[256];
  // every block recomputes the sum of the non-zero entries (65536 words: cheap) -- only block 0 needs it
  const uint32_t r = blockIdx.x * 256u + threadIdx.x;
  if (blockIdx.x == 0) {
    uint32_t s = 0;
    for (uint32_t i = threadIdx.x; i < 65536u; i += 256) s += i ? hist[i] : 0u;
    part[threadIdx.x] = s;
    __syncthreads();
    for (uint32_t k = 128; k; k >>= 1) {
      if (threadIdx.x < k) part[threadIdx.x] += part[threadIdx.x + k];
      __syncthreads();
    }
  }
  if (r >= n) return;
  uint32_t v = r < 65536u ? hist[r] : 0u;
  if (r == 0) v = total - part[0];
  column[r] = enc(v);
}

// the row's fractions, accumulators [a0, a1): chain links leave their running sum within the row in the ACCUM columns and the
// row total in `row_total`; an accumulator with a public total leaves its term in its own scan buffer
__global__ __launch_bounds__(256) void logup_term_kernel(uint32_t* __restrict__ accum, uint32_t* __restrict__ row_total, uint32_t* __restrict__ own_terms,
                                                         const uint32_t* __restrict__ tape, const uint32_t* const* __restrict__ cols, const Fp4* __restrict__ ch,
                                                         uint32_t a0, uint32_t a1, uint32_t n_chain, uint32_t po2) {
  const uint32_t r = blockIdx.x * 256u + threadIdx.x, n = 1u << po2;
  if (r >= n) return;
  uint32_t at = 0;
  Fp4 run = fp4_zero();
  for (uint32_t j = a0; j < a1; j++) {
    Fp4 d[4];
    uint32_t num[4];
    for (uint32_t f = 0; f < 4; f++) {
      at++;  // table
      num[f] = eval_form(tape, at, cols, r);
      const uint32_t n_parts = tape[at++];
      Fp4 den = fp4_zero();
      for (uint32_t q = 0; q < n_parts; q++) {
        const uint32_t ci = tape[at++];
        const uint32_t v = eval_form(tape, at, cols, r);
        if (ci == 0) den.e[0] = add(den.e[0], v);
        else den = den + scale(ch[ci], v);
      }
      d[f] = den;
    }
    const Fp4 d01 = d[0] * d[1], d23 = d[2] * d[3];
    const Fp4 top = (scale(d[1], num[0]) + scale(d[0], num[1])) * d23 + (scale(d[3], num[2]) + scale(d[2], num[3])) * d01;
    const Fp4 term = top * fp4_inv(d01 * d23);
    if (j < n_chain) {
      run = run + term;
      for (uint32_t i = 0; i < 4; i++) accum[((size_t)(4 * j + i) << po2) + r] = run.e[i];
    } else {
      *(uint4*)(own_terms + 4 * ((size_t)(j - n_chain) * n + r)) = make_uint4(term.e[0], term.e[1], term.e[2], term.e[3]);
    }
  }
  if (a0 < n_chain) *(uint4*)(row_total + 4 * (size_t)r) = make_uint4(run.e[0], run.e[1], run.e[2], run.e[3]);
}
// chain links: add the sum of all earlier rows (inclusive scan of the row totals, one row back)
__global__ void logup_chain_kernel(uint32_t* __restrict__ accum, const uint32_t* __restrict__ scanned, uint32_t n_cols, uint32_t po2) {
  const uint32_t r = blockIdx.x * 256u + threadIdx.x, col = blockIdx.y;
  if (r == 0 || r >= (1u << po2) || col >= n_cols) return;
  uint32_t* cell = accum + ((size_t)col << po2) + r;
  *cell = add(*cell, scanned[4 * (size_t)(r - 1) + (col & 3u)]);
}
__global__ void logup_unpack_kernel(uint32_t* __restrict__ cols, const uint32_t* __restrict__ scanned, uint32_t po2) {
  const uint32_t r = blockIdx.x * 256u + threadIdx.x;
  if (r >= (1u << po2)) return;
  const uint4 v = *(const uint4*)(scanned + 4 * (size_t)r);
  cols[r] = v.x; cols[((size_t)1 << po2) + r] = v.y; cols[((size_t)2 << po2) + r] = v.z; cols[((size_t)3 << po2) + r] = v.w;
}
}  // namespace

static const char* launch_check(const char* what) {
  hipError_t e = hipGetLastError();
  R0H_REQUIRE(e == hipSuccess, "%s: %s", what, hipGetErrorString(e));
  return nullptr;
}

// standalone accumulators: terms -> running sums; totals_out (host, 4 words each) if wanted, ACCUM columns if `accum`
static const char* own_accumulators(r0h_ctx* ctx, const r0h_circuit* c, uint32_t po2, const Tape& t, const DeviceTape& d, r0h_buf* accum, uint32_t* totals_out) {
  const uint32_t n = 1u << po2, n_chain = c->logup.n_chain, n_acc = (uint32_t)c->logup.accs.size(), n_own = n_acc - n_chain;
  if (!n_own) return nullptr;
  r0h_buf* terms = nullptr;
  R0H_TRY(buf_alloc_pooled(ctx, (size_t)n_own * n * 16, &terms));
  struct Free { r0h_buf* b; ~Free() { r0h_buf_free(b); } } guard{terms};
  hipLaunchKernelGGL(logup_term_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, accum ? u32(accum) : nullptr, (uint32_t*)nullptr, u32(terms),
                     d.words + t.acc_begin[n_chain], d.cols, d.ch, n_chain, n_acc, n_chain, po2);
  R0H_TRY(launch_check("logup_term_kernel"));
  for (uint32_t k = 0; k < n_own; k++) {
    r0h_buf view = *terms;
    view.ptr = (char*)terms->ptr + (size_t)k * n * 16;
    view.bytes = (size_t)n * 16;
    R0H_TRY(r0h_prefix_sums(ctx, &view, n));
    if (accum) {
      hipLaunchKernelGGL(logup_unpack_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, u32(accum) + ((size_t)(4 * (n_chain + k)) << po2), (const uint32_t*)view.ptr, po2);
      R0H_TRY(launch_check("logup_unpack_kernel"));
    }
    if (totals_out) R0H_TRY(r0h_buf_d2h(ctx, &view, (size_t)(n - 1) * 16, totals_out + 4 * k, 16));
  }
  return nullptr;
}

const char* logup_accum(r0h_ctx* ctx, const r0h_circuit* c, uint32_t po2, const r0h_buf* code, const r0h_buf* data, const uint32_t* global, const uint32_t* mix, r0h_buf* accum) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(ctx && c && data && accum, "r0h_accum: NULL argument");
  R0H_REQUIRE(po2 >= 4 && po2 <= R0H_MAX_PO2, "r0h_accum: po2 %u outside [4, %u]", po2, R0H_MAX_PO2);
  const uint32_t n = 1u << po2, n_chain = c->logup.n_chain;
  R0H_REQUIRE(((size_t)c->group_size[R0H_GROUP_DATA] << po2) * 4 <= data->bytes && ((size_t)c->group_size[R0H_GROUP_ACCUM] << po2) * 4 <= accum->bytes &&
                  (!code || ((size_t)c->group_size[R0H_GROUP_CODE] << po2) * 4 <= code->bytes),
              "r0h_accum: buffers too small for 2^%u rows", po2);
  for (uint32_t i = 0; i < c->n_mix; i++) R0H_REQUIRE(mix && mix[i] < P, "r0h_accum: mix[%u] missing or not canonical", i);
  for (uint32_t i = 0; i < c->n_global; i++) R0H_REQUIRE(!global || global[i] < P, "r0h_accum: global[%u] not canonical", i);
  Tape t;
  R0H_TRY(build_tape(c, po2, code, data, global, mix, &t));
  DeviceTape d;
  R0H_TRY(upload_tape(ctx, t, &d));
  KScope ks(ctx, "logup_accum", ((double)t.cols.size() + 2.0 * c->group_size[R0H_GROUP_ACCUM]) * n * 4);
  r0h_buf* totals = nullptr;
  R0H_TRY(buf_alloc_pooled(ctx, (size_t)n * 16, &totals));
  struct Free { r0h_buf* b; ~Free() { r0h_buf_free(b); } } guard{totals};
  hipLaunchKernelGGL(logup_term_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, u32(accum), u32(totals), (uint32_t*)nullptr, d.words, d.cols, d.ch, 0u, n_chain, n_chain, po2);
  R0H_TRY(launch_check("logup_term_kernel"));
  R0H_TRY(r0h_prefix_sums(ctx, totals, n));
  hipLaunchKernelGGL(logup_chain_kernel, dim3((n + 255) / 256, 4 * n_chain), dim3(256), 0, ctx->stream, u32(accum), u32(totals), 4 * n_chain, po2);
  R0H_TRY(launch_check("logup_chain_kernel"));
  R0H_TRY(own_accumulators(ctx, c, po2, t, d, accum, nullptr));
  R0H_TRY_HIP(hipStreamSynchronize(ctx->stream));  // the tape goes back to the pool
  return nullptr;
  R0H_GUARD_END
}

}  // namespace r0h

using namespace r0h;

extern "C" {

// Fill the multiplicity columns of `data` from the lookups its rows make (before DATA is committed).
const char* r0h_logup_multiplicities(r0h_ctx* ctx, const r0h_circuit* c, uint32_t po2, r0h_buf* data, const uint32_t* global) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(ctx && c && data, "r0h_logup_multiplicities: NULL argument");
  if (c->logup.tables.empty()) return nullptr;
  R0H_REQUIRE(po2 >= 16 && po2 <= R0H_MAX_PO2, "r0h_logup_multiplicities: the tables have 2^16 rows: po2 %u outside [16, %u]", po2, R0H_MAX_PO2);
  R0H_REQUIRE(((size_t)c->group_size[R0H_GROUP_DATA] << po2) * 4 <= data->bytes, "r0h_logup_multiplicities: the DATA buffer is too small for 2^%u rows", po2);
  R0H_REQUIRE(c->logup.tables.size() <= 2, "r0h_logup_multiplicities: at most two tables");
  const uint32_t n = 1u << po2, n_tables = (uint32_t)c->logup.tables.size();
  // only the lookups are on this tape: chain links in order, challenges unused (their indices are skipped)
  Tape t;
  std::vector<uint32_t> dummy_mix(c->n_mix, 0), dummy_global(c->n_global, 0);
  R0H_TRY(build_tape(c, po2, nullptr, data, global ? global : dummy_global.data(), dummy_mix.data(), &t, true));  // the lookups alone (they read DATA only)
  DeviceTape d;
  R0H_TRY(upload_tape(ctx, t, &d));
  std::vector<uint64_t> lookups(n_tables, 0);
  for (uint32_t j = 0; j < c->logup.n_chain; j++)
    for (const LogupFraction& f : c->logup.accs[j].fr)
      if (f.table && f.table <= n_tables) lookups[f.table - 1]++;
  r0h_buf* hist = nullptr;
  R0H_TRY(buf_alloc_pooled(ctx, (size_t)n_tables * 65536 * 4, &hist));
  struct Free { r0h_buf* b; ~Free() { r0h_buf_free(b); } } guard{hist};
  R0H_TRY_HIP(hipMemsetAsync(hist->ptr, 0, (size_t)n_tables * 65536 * 4, ctx->stream));
  KScope ks(ctx, "logup_multiplicities", (double)t.cols.size() * n * 4);
  hipLaunchKernelGGL(logup_count_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, u32(hist), d.words, (uint32_t)t.words.size(), d.cols, n, n_tables);
  R0H_TRY(launch_check("logup_count_kernel"));
  for (uint32_t k = 0; k < n_tables; k++) {
    const LogupTable& tb = c->logup.tables[k];
    R0H_REQUIRE(tb.kind == k + 1, "r0h_logup_multiplicities: table %u is not of kind %u", k, k + 1);
    hipLaunchKernelGGL(logup_mult_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, u32(data) + ((size_t)tb.data_col << po2), u32(hist) + (size_t)k * 65536,
                       (uint32_t)(lookups[k] * n), n);
    R0H_TRY(launch_check("logup_mult_kernel"));
  }
  R0H_TRY_HIP(hipStreamSynchronize(ctx->stream));
  return nullptr;
  R0H_GUARD_END
}

// The totals of the accumulators that run alone (their challenges are public inputs, so they can be had before the mix is drawn):
// global_io[final .. final + 4) of each is overwritten with its total over the rows.
const char* r0h_logup_totals(r0h_ctx* ctx, const r0h_circuit* c, uint32_t po2, const r0h_buf* code, const r0h_buf* data, uint32_t* global_io) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(ctx && c && data && global_io, "r0h_logup_totals: NULL argument");
  R0H_REQUIRE(po2 >= 4 && po2 <= R0H_MAX_PO2, "r0h_logup_totals: po2 %u outside [4, %u]", po2, R0H_MAX_PO2);
  const uint32_t n_chain = c->logup.n_chain, n_acc = (uint32_t)c->logup.accs.size();
  if (n_acc == n_chain) return nullptr;
  for (uint32_t j = n_chain; j < n_acc; j++)
    for (const LogupFraction& f : c->logup.accs[j].fr)
      for (const LogupPart& q : f.parts) R0H_REQUIRE(q.ch_kind != 1, "r0h_logup_totals: accumulator %u takes a challenge from the mix", j);
  Tape t;
  std::vector<uint32_t> dummy_mix(c->n_mix, 0);
  R0H_TRY(build_tape(c, po2, code ? code : data, data, global_io, dummy_mix.data(), &t));
  DeviceTape d;
  R0H_TRY(upload_tape(ctx, t, &d));
  std::vector<uint32_t> totals(4 * (size_t)(n_acc - n_chain));
  R0H_TRY(own_accumulators(ctx, c, po2, t, d, nullptr, totals.data()));
  for (uint32_t j = n_chain; j < n_acc; j++) memcpy(global_io + c->logup.accs[j].final_global, totals.data() + 4 * (size_t)(j - n_chain), 16);
  return nullptr;
  R0H_GUARD_END
}

}  // extern "C"
