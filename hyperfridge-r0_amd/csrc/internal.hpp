// Internal definitions shared by the translation units of libr0hip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/r0hip.h"
#include "fp.hpp"

namespace r0h {

const char* make_error(const char* fmt, ...);

#define R0H_TRY_HIP(expr)                                                                              \
  do {                                                                                                 \
    hipError_t e__ = (expr);                                                                           \
    if (e__ != hipSuccess) return r0h::make_error("%s:%d: %s: %s", __FILE__, __LINE__, #expr, hipGetErrorString(e__)); \
  } while (0)
#define R0H_TRY(expr)                \
  do {                               \
    const char* m__ = (expr);        \
    if (m__) return m__;             \
  } while (0)
#define R0H_REQUIRE(cond, ...)                        \
  do {                                                \
    if (!(cond)) return r0h::make_error(__VA_ARGS__); \
  } while (0)
#define R0H_GUARD_BEGIN try {
#define R0H_GUARD_END \
  }                   \
  catch (const std::exception& ex) { return r0h::make_error("exception: %s", ex.what()); } \
  catch (...) { return r0h::make_error("unknown exception"); }

constexpr int P2_CELLS = 24, P2_RATE = 16, P2_OUT = 8, P2_HALF_FULL = 4, P2_PARTIAL = 21, P2_ROUNDS = 29;
constexpr int P2_PART_SIGMA_WORDS = (P2_PARTIAL - 1) * (P2_CELLS - 1) + (P2_PARTIAL - 1) * P2_PARTIAL / 2;  // sum_{r=1}^{20} (23 + r) = 670
constexpr uint32_t TW_BITS = 11;            // two-level twiddle tables of 2^11 entries each
constexpr uint32_t TW_SIZE = 1u << TW_BITS;
constexpr uint32_t TWL_BITS = 13;           // in-chunk twiddle table: ROU[13]^i, i < 2^12 (chunks of up to 2^13 words)
constexpr uint32_t TW_TOP = 22;             // the two-level tables hold powers of ROU[22]: every transform of the default segment size
constexpr uint32_t TWB_BITS = 13;           // second table pair for domains above 2^22: powers of ROU[26], 2^13 entries each
constexpr uint32_t TWB_SIZE = 1u << TWB_BITS;
constexpr uint32_t MAX_DOMAIN_PO2 = 26;     // 2^R0H_MAX_PO2 rows x INV_RATE
// Per context: one po2 = 20 segment parks about 8 GiB, one of 2^24 rows about 130 GiB.  What exceeds the limit is released on
// free; an allocation the device refuses drains the context's own pool and is tried once more (buf_alloc_pooled).
constexpr size_t POOL_LIMIT = (size_t)192 << 30;

// Device-resident Poseidon2 tables (Montgomery form).
struct P2Consts {
  uint32_t rc_full[2 * P2_HALF_FULL][P2_CELLS];
  uint32_t rc_partial[P2_PARTIAL];
  uint32_t diag[P2_CELLS];        // Montgomery form of (mu_i - 1)
  uint32_t diag_canon[P2_CELLS];  // canonical (mu_i - 1) and its Shoup companion floor(w 2^32 / p): constant products
  uint32_t diag_shoup[P2_CELLS];
  // Partial rounds with the linear layer of lanes 1..23 deferred (poseidon2_device.hpp): with D_i = mu_i - 1 and
  // kappa_k = sum_{i>=1} D_i^k, all in Montgomery form,
  //   part_sigma: for r = 1..20 the row [D_1^r .. D_23^r, kappa_{r-1}, kappa_{r-2}, .., kappa_0]   (23 + r words each)
  //   part_final: for i = 1..23 the row [D_i^21, D_i^20, .., D_i^0]
  uint32_t part_sigma[P2_PART_SIGMA_WORDS];
  uint32_t part_final[(P2_CELLS - 1) * (P2_PARTIAL + 1)];
};

// Optional per-kernel timing (HIP events on the context's stream around every launch of a named kernel family).
struct KTimer {
  std::vector<hipEvent_t> ev;  // start/stop pairs
  size_t used = 0;
  double alg_bytes = 0;        // algorithmic HBM bytes of the recorded launches
};

struct Profile {
  std::vector<const char*> names;
  std::vector<float> ms;
  std::vector<hipEvent_t> events;  // events[i], events[i+1] bracket phase i
};

}  // namespace r0h

struct r0h_ctx {
  int device = 0;
  int refs = 1;  // the handle itself + every live buffer / circuit: teardown happens when the last one goes

  hipStream_t stream = nullptr;
  // twiddles: tw_lo[d][i] = w^i, tw_hi[d][i] = w^(i*2^11) with w = ROU_{FWD,REV}[22]; d = 0 forward, 1 inverse
  uint32_t* tw_lo[2] = {nullptr, nullptr};
  uint32_t* tw_hi[2] = {nullptr, nullptr};
  // local table: tw12[d][i] = ROU[12]^i for i < 2048
  uint32_t* tw12[2] = {nullptr, nullptr};
  // the same for w = ROU_{FWD,REV}[26] with 2^13 entries per level: outer pass of transforms above 2^22 points
  uint32_t* twb_lo[2] = {nullptr, nullptr};
  uint32_t* twb_hi[2] = {nullptr, nullptr};
  uint32_t* pow3_lo = nullptr;  // 3^i, i < 2^11
  uint32_t* pow3_hi = nullptr;  // 3^(i*2^11), i < 2^11 (covers exponents < 2^22)
  uint32_t* pow3_top = nullptr; // 3^(i*2^22), i < 16 (exponents up to 2^26)
  r0h::P2Consts* p2 = nullptr;  // device
  r0h::P2Consts p2_host;
  void* scratch = nullptr;      // small device scratch for scans / partial sums
  size_t scratch_bytes = 0;
  void* pinned = nullptr;       // pinned host staging ring for small parameter uploads
  size_t pinned_bytes = 0, pinned_off = 0;
  std::multimap<size_t, void*> pool;  // cached device allocations of the sequencer, by size (stream-ordered reuse)
  size_t pool_bytes = 0;              // bytes parked in the pool; above POOL_LIMIT blocks are released instead
  r0h::Profile prof;
  r0h_session_stats session = {0, 0, 0, 0, 0, 0, 0};  // stage timing of the last r0h_prove_elf
  const r0h_circuit* image_circuit = nullptr;  // r0h_ctx_set_image_circuit: sessions on this context attach an image proof to their receipts
  uint64_t session_resident_limit = 0;  // r0h_ctx_set_session_resident_limit (0: an eighth of the device's memory)
  void* session_rows = nullptr;   // session.cpp: the preflight row buffers of r0h_prove_elf, page-locked, kept from one call to the next (session_rows_free)
  std::vector<r0h_ctx*> helpers;  // further contexts of the same device, made on demand by r0h_prove_elf for its extra prover lanes; they go with this one
  bool ktime_on = false;
  std::map<std::string, r0h::KTimer> ktimers;
};

struct r0h_buf {
  r0h_ctx* ctx = nullptr;
  void* ptr = nullptr;
  size_t bytes = 0;
  r0h_buf* parent = nullptr;  // slices keep their parent alive
  int refs = 1;
  bool owned = true;
  bool pooled = false;  // memory returns to the context's pool instead of hipFree
};

namespace r0h {
inline uint32_t* u32(const r0h_buf* b) { return (uint32_t*)b->ptr; }
// RAII bracket around the launches of one kernel family; no-op unless r0h_kernel_timing(ctx, 1) was called
struct KScope {
  r0h_ctx* ctx;
  KTimer* t = nullptr;
  KScope(r0h_ctx* c, const char* name, double alg_bytes) : ctx(c) {
    // every operation opens a scope before it launches: also the place where the calling thread is pointed at the context's
    // device (a host thread may drive contexts on several devices; allocations and function attributes go to the current one)
    (void)hipSetDevice(c->device);
    if (!c->ktime_on) return;
    t = &c->ktimers[name];
    if (t->ev.size() < t->used + 2) {
      hipEvent_t a, b;
      (void)hipEventCreate(&a);
      (void)hipEventCreate(&b);
      t->ev.push_back(a);
      t->ev.push_back(b);
    }
    t->alg_bytes += alg_bytes;
    (void)hipEventRecord(t->ev[t->used], c->stream);
  }
  ~KScope() {
    if (!t) return;
    (void)hipEventRecord(t->ev[t->used + 1], ctx->stream);
    t->used += 2;
  }
};
const char* ensure_scratch(r0h_ctx* ctx, size_t bytes);
// stream-ordered upload of a small host array through the pinned ring (the caller's memory may die on return)
const char* stage_h2d(r0h_ctx* ctx, void* dst_device, const void* src_host, size_t bytes);
// device buffer from the context's pool: no hipMalloc / hipFree (and no implicit device sync) in steady state
const char* buf_alloc_pooled(r0h_ctx* ctx, size_t bytes, r0h_buf** out);
// inverse NTT with the coset shift f(x) -> f(3x) optionally fused into its last pass (sequencer path)
const char* interpolate_ntt(r0h_ctx* ctx, r0h_buf* io, const r0h_buf* src, uint32_t count, uint32_t po2, bool zk_shift);  // src may be io
// batch_evaluate_any over coefficients stored in natural or bit-reversed order (the sequencer keeps them bit-reversed)
const char* evaluate_any(r0h_ctx* ctx, const r0h_buf* coeffs, uint32_t po2, const uint32_t* which, const uint32_t* xs, uint32_t n_eval,
                         r0h_buf* out, bool bitrev_coeffs);
// in-place bit reversal of `count` columns of 2^po2 extension elements (16-byte units)
const char* bit_reverse_ext(r0h_ctx* ctx, r0h_buf* io, uint32_t count, uint32_t po2);
// per-device kernel attributes of the NTT family (LDS limits); called by r0h_ctx_create with the device current
const char* ntt_init_device();
// batched synthetic division (DEEP step): job j divides polynomial poly_idx[j] of `polys` by (x - points[4j..]); one read-back
const char* poly_divide_batch(r0h_ctx* ctx, r0h_buf* polys, uint32_t n, const uint32_t* poly_idx, const uint32_t* points, uint32_t n_jobs, uint32_t* remainders_host);
// rv32im.cpp: the preflight rows of segment i are moved out of the machine (the session proves them while the guest runs on)
void vm_take_trace(r0h_vm* vm, size_t i, std::vector<r0h_preflight_row>& rows, std::vector<r0h_preflight_bound>& bounds);
void vm_recycle_trace(r0h_vm* vm, std::vector<r0h_preflight_row>& rows, std::vector<r0h_preflight_bound>& bounds);
void vm_take_spares(r0h_vm* vm, std::vector<std::vector<r0h_preflight_row>>& rows, std::vector<std::vector<r0h_preflight_bound>>& bounds);
void session_rows_free(r0h_ctx* ctx);  // session.cpp: unpins and frees the row buffers a context kept (ctx teardown, device still alive)
// the k-th helper context of `ctx` (same device, same Poseidon2 table), created on first use and kept until ctx goes
const char* ctx_helper(r0h_ctx* ctx, size_t k, r0h_ctx** out);
void ctx_retain(r0h_ctx* ctx);
void ctx_release(r0h_ctx* ctx);
// host Poseidon2 (transcript only): permutation over 24 Montgomery words with the context's table
void p2_mix_host(const P2Consts& k, uint32_t* cells);
// table from canonical round constants [29][24] and canonical (mu_i - 1) [24]; the compiled-in risc0 table
void fill_p2(P2Consts& k, const uint32_t* rc, const uint32_t* diag_m1);
void p2_default_host(P2Consts& k);
void p2_hash_elems_host(const P2Consts& k, const uint32_t* elems, size_t n, uint32_t digest[8]);
void p2_sponge_rows_host(const P2Consts& k, const uint32_t* words, size_t n_words, uint32_t* cols, size_t stride, size_t* rows_used);
}  // namespace r0h
