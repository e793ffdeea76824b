// Context, device buffers, error strings, constant tables.
// Replaces the host-side glue of risc0-zkp 3.0.4 hal/{cuda,metal}.rs (`alloc_elem`, `copy_from_*`, buffer slicing) and
// risc0-sys 1.5.0's C-string error convention.
#include <stdarg.h>

#include <exception>

#include "../../include/r0hip_poseidon2_consts.h"
#include "internal.hpp"

namespace r0h {

const char* make_error(const char* fmt, ...) {
  char tmp[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(tmp, sizeof tmp, fmt, ap);
  va_end(ap);
  char* out = (char*)malloc(strlen(tmp) + 1);
  if (!out) return "r0hip: out of memory while formatting an error";
  strcpy(out, tmp);
  return out;
}

void fill_p2(P2Consts& k, const uint32_t* rc, const uint32_t* diag) {
  int fr = 0, pr = 0;
  for (int r = 0; r < P2_ROUNDS; r++) {
    bool full = r < P2_HALF_FULL || r >= P2_HALF_FULL + P2_PARTIAL;
    if (full) {
      for (int i = 0; i < P2_CELLS; i++) k.rc_full[fr][i] = enc(rc[r * P2_CELLS + i]);
      fr++;
    } else {
      k.rc_partial[pr++] = enc(rc[r * P2_CELLS]);
    }
  }
  for (int i = 0; i < P2_CELLS; i++) {
    k.diag[i] = enc(diag[i]);
    k.diag_canon[i] = diag[i];
    k.diag_shoup[i] = shoup_companion(diag[i]);
  }
  // tables of the deferred partial rounds: powers of the diagonal and their lane sums (canonical arithmetic, then encoded)
  uint32_t pw[P2_PARTIAL + 1][P2_CELLS], kappa[P2_PARTIAL + 1];
  for (int e = 0; e <= P2_PARTIAL; e++) {
    uint64_t ks = 0;
    for (int i = 1; i < P2_CELLS; i++) {
      pw[e][i] = e == 0 ? 1u : (uint32_t)((uint64_t)pw[e - 1][i] * diag[i] % P);
      ks += pw[e][i];
    }
    kappa[e] = (uint32_t)(ks % P);
  }
  uint32_t* w = k.part_sigma;
  for (int r = 1; r < P2_PARTIAL; r++) {
    for (int i = 1; i < P2_CELLS; i++) *w++ = enc(pw[r][i]);
    for (int j = 0; j < r; j++) *w++ = enc(kappa[r - 1 - j]);
  }
  w = k.part_final;
  for (int i = 1; i < P2_CELLS; i++)
    for (int e = P2_PARTIAL; e >= 0; e--) *w++ = enc(pw[e][i]);
}
void p2_default_host(P2Consts& k) { fill_p2(k, R0H_P2_ROUND_CONSTANTS, R0H_P2_INT_DIAG_M1); }

static inline uint32_t sbox7(uint32_t x) {
  uint32_t x2 = mul(x, x), x4 = mul(x2, x2);
  return mul(mul(x4, x2), x);
}
static void m_ext_host(uint32_t* c) {
  uint32_t col[4] = {0, 0, 0, 0};
  for (int k = 0; k < P2_CELLS; k += 4) {
    uint32_t a = c[k], b = c[k + 1], d = c[k + 2], e = c[k + 3];
    uint32_t t0 = add(a, b), t1 = add(d, e);
    uint32_t t2 = add(add(b, b), t1), t3 = add(add(e, e), t0);
    uint32_t t4 = add(add(add(t1, t1), add(t1, t1)), t3), t5 = add(add(add(t0, t0), add(t0, t0)), t2);
    c[k] = add(t3, t5); c[k + 1] = t5; c[k + 2] = add(t2, t4); c[k + 3] = t4;
    for (int j = 0; j < 4; j++) col[j] = add(col[j], c[k + j]);
  }
  for (int i = 0; i < P2_CELLS; i++) c[i] = add(c[i], col[i & 3]);
}
void p2_mix_host(const P2Consts& k, uint32_t* c) {
  m_ext_host(c);
  for (int r = 0; r < P2_HALF_FULL; r++) {
    for (int i = 0; i < P2_CELLS; i++) c[i] = sbox7(add(c[i], k.rc_full[r][i]));
    m_ext_host(c);
  }
  for (int r = 0; r < P2_PARTIAL; r++) {
    c[0] = sbox7(add(c[0], k.rc_partial[r]));
    uint32_t sum = 0;
    for (int i = 0; i < P2_CELLS; i++) sum = add(sum, c[i]);
    for (int i = 0; i < P2_CELLS; i++) c[i] = add(sum, mul(k.diag[i], c[i]));
  }
  for (int r = P2_HALF_FULL; r < 2 * P2_HALF_FULL; r++) {
    for (int i = 0; i < P2_CELLS; i++) c[i] = sbox7(add(c[i], k.rc_full[r][i]));
    m_ext_host(c);
  }
}
// The sponge below (p2_hash_elems_host) laid out row by row for the recursion circuit's in-circuit hash (include/r0hip_circuit.h SPONGE,
// tools/sponge_component.py): 30 rows per permutation -- absorb + external layer, then one round per row -- in 65 columns of `stride`
// words: st[24] the state after the row's step, aux[24] the cubes (x + rc)^3 of the lanes the round's S-box touches, in[16] the words
// absorbed on a permutation's first row, act = 1.  Only rows [0, *rows_used) are written; the caller clears the rest.
void p2_sponge_rows_host(const P2Consts& k, const uint32_t* words, size_t n_words, uint32_t* cols, size_t stride, size_t* rows_used) {
  constexpr int ROWS = 1 + 2 * P2_HALF_FULL + P2_PARTIAL;  // one permutation: the absorbing row, then a row per round
  const size_t n_perm = n_words ? (n_words + P2_RATE - 1) / P2_RATE : 1;
  uint32_t st[P2_CELLS] = {0};
  // a permutation's rows are made in a small block [column][row of the permutation] and go out column by column: 65 runs of 30
  // consecutive words instead of 1,950 single words a stride apart
  uint32_t blk[65][ROWS];
  for (size_t q = 0; q < n_perm; q++) {
    memset(blk, 0, sizeof blk);
    int row = 0;
    auto put_state = [&] {
      for (int j = 0; j < P2_CELLS; j++) blk[j][row] = st[j];
      blk[64][row] = ONE;
    };
    for (size_t j = 0; j < P2_RATE; j++) {
      st[j] = q * P2_RATE + j < n_words ? words[q * P2_RATE + j] : 0u;
      blk[48 + j][0] = st[j];
    }
    m_ext_host(st);
    put_state();
    row++;
    for (int r = 0; r < 2 * P2_HALF_FULL + P2_PARTIAL; r++, row++) {
      const bool full = r < P2_HALF_FULL || r >= P2_HALF_FULL + P2_PARTIAL;
      const uint32_t* rc = full ? k.rc_full[r < P2_HALF_FULL ? r : r - P2_PARTIAL] : &k.rc_partial[r - P2_HALF_FULL];
      for (int j = 0; j < (full ? P2_CELLS : 1); j++) {
        const uint32_t t = add(st[j], rc[j]), cube = mul(mul(t, t), t);
        st[j] = mul(mul(cube, cube), t);
        blk[24 + j][row] = cube;
      }
      if (full) m_ext_host(st);
      else {
        uint32_t sum = 0;
        for (int i = 0; i < P2_CELLS; i++) sum = add(sum, st[i]);
        for (int i = 0; i < P2_CELLS; i++) st[i] = add(sum, mul(k.diag[i], st[i]));
      }
      put_state();
    }
    for (int c = 0; c < 65; c++) memcpy(cols + (size_t)c * stride + q * ROWS, blk[c], sizeof blk[c]);
  }
  *rows_used = n_perm * ROWS;
}
void p2_hash_elems_host(const P2Consts& k, const uint32_t* elems, size_t n, uint32_t digest[8]) {
  uint32_t st[P2_CELLS] = {0};
  size_t used = 0;
  for (size_t i = 0; i < n; i++) {
    st[used++] = elems[i];
    if (used == P2_RATE) { p2_mix_host(k, st); used = 0; }
  }
  if (used != 0 || n == 0) {
    for (size_t i = used; i < P2_RATE; i++) st[i] = 0;
    p2_mix_host(k, st);
  }
  memcpy(digest, st, 32);
}

const char* ensure_scratch(r0h_ctx* ctx, size_t bytes) {
  if (ctx->scratch_bytes >= bytes) return nullptr;
  if (ctx->scratch) {
    R0H_TRY_HIP(hipStreamSynchronize(ctx->stream));
    R0H_TRY_HIP(hipFree(ctx->scratch));
    ctx->scratch = nullptr;
    ctx->scratch_bytes = 0;
  }
  R0H_TRY_HIP(hipMalloc(&ctx->scratch, bytes));
  ctx->scratch_bytes = bytes;
  return nullptr;
}

const char* stage_h2d(r0h_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (!bytes) return nullptr;
  if (bytes > ctx->pinned_bytes / 2) {  // too large for the ring: plain blocking copy
    R0H_TRY_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    R0H_TRY_HIP(hipStreamSynchronize(ctx->stream));
    return nullptr;
  }
  size_t need = (bytes + 255) & ~(size_t)255;
  if (ctx->pinned_off + need > ctx->pinned_bytes) {
    R0H_TRY_HIP(hipStreamSynchronize(ctx->stream));  // every earlier ring slot has been consumed
    ctx->pinned_off = 0;
  }
  char* slot = (char*)ctx->pinned + ctx->pinned_off;
  memcpy(slot, src, bytes);
  ctx->pinned_off += need;
  R0H_TRY_HIP(hipMemcpyAsync(dst, slot, bytes, hipMemcpyHostToDevice, ctx->stream));
  return nullptr;
}

const char* buf_alloc_pooled(r0h_ctx* ctx, size_t bytes, r0h_buf** out) {
  size_t sz = bytes ? (bytes + 255) & ~(size_t)255 : 256;
  r0h_buf* b = new r0h_buf();
  b->ctx = ctx; b->bytes = bytes; b->pooled = true;
  auto it = ctx->pool.find(sz);
  if (it != ctx->pool.end()) {
    b->ptr = it->second;
    ctx->pool.erase(it);
    ctx->pool_bytes -= sz;
  } else {
    hipError_t e = hipMalloc(&b->ptr, sz);
    if (e != hipSuccess && !ctx->pool.empty()) {
      // out of memory with blocks of other sizes parked: give them back and try once more
      (void)hipGetLastError();
      (void)hipStreamSynchronize(ctx->stream);
      for (auto& kv : ctx->pool) (void)hipFree(kv.second);
      ctx->pool.clear();
      ctx->pool_bytes = 0;
      e = hipMalloc(&b->ptr, sz);
    }
    if (e != hipSuccess) {
      (void)hipGetLastError();
      delete b;
      return make_error("device allocation of %zu bytes failed: %s", sz, hipGetErrorString(e));
    }
  }
  ctx_retain(ctx);
  *out = b;
  return nullptr;
}

void ctx_retain(r0h_ctx* ctx) { ctx->refs++; }
const char* ctx_helper(r0h_ctx* ctx, size_t k, r0h_ctx** out) {
  while (ctx->helpers.size() <= k) {
    r0h_ctx* h = nullptr;
    R0H_TRY(r0h_ctx_create(ctx->device, &h));
    ctx->helpers.push_back(h);
  }
  r0h_ctx* h = ctx->helpers[k];
  if (memcmp(&h->p2_host, &ctx->p2_host, sizeof(P2Consts)) != 0) {  // r0h_poseidon2_set_consts on the owner since the helper was made
    h->p2_host = ctx->p2_host;
    R0H_TRY_HIP(hipSetDevice(h->device));
    R0H_TRY_HIP(hipStreamSynchronize(h->stream));
    R0H_TRY_HIP(hipMemcpy(h->p2, &h->p2_host, sizeof(P2Consts), hipMemcpyHostToDevice));
  }
  *out = h;
  return nullptr;
}
void ctx_release(r0h_ctx* ctx) {
  if (--ctx->refs > 0) return;
  session_rows_free(ctx);
  for (r0h_ctx* h : ctx->helpers) r0h_ctx_destroy(h);
  ctx->helpers.clear();
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  for (int d = 0; d < 2; d++) { (void)hipFree(ctx->tw_lo[d]); (void)hipFree(ctx->tw_hi[d]); (void)hipFree(ctx->tw12[d]); }
  for (int d = 0; d < 2; d++) { (void)hipFree(ctx->twb_lo[d]); (void)hipFree(ctx->twb_hi[d]); }
  (void)hipFree(ctx->pow3_lo); (void)hipFree(ctx->pow3_hi); (void)hipFree(ctx->pow3_top); (void)hipFree(ctx->p2); (void)hipFree(ctx->scratch);
  (void)hipHostFree(ctx->pinned);
  for (auto& kv : ctx->pool) (void)hipFree(kv.second);
  for (hipEvent_t e : ctx->prof.events) (void)hipEventDestroy(e);
  for (auto& kv : ctx->ktimers)
    for (hipEvent_t e : kv.second.ev) (void)hipEventDestroy(e);
  (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

static const char* upload_pow_table(uint32_t** dst, uint32_t base, uint32_t n) {
  std::vector<uint32_t> t(n);
  uint32_t cur = ONE;
  for (uint32_t i = 0; i < n; i++) { t[i] = cur; cur = mul(cur, base); }
  R0H_TRY_HIP(hipMalloc((void**)dst, n * 4));
  R0H_TRY_HIP(hipMemcpy(*dst, t.data(), n * 4, hipMemcpyHostToDevice));
  return nullptr;
}

}  // namespace r0h

using namespace r0h;

extern "C" {

void r0h_free_error(const char* msg) { free((void*)msg); }
const char* r0h_version(void) { return "r0hip 0.1 (gfx950)"; }

const char* r0h_ctx_create(int device, r0h_ctx** out) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(out, "r0h_ctx_create: out is NULL");
  int n = 0;
  R0H_TRY_HIP(hipGetDeviceCount(&n));
  R0H_REQUIRE(device >= 0 && device < n, "r0h_ctx_create: device %d not present (%d visible); no CPU fallback exists", device, n);
  R0H_TRY_HIP(hipSetDevice(device));
  R0H_TRY(ntt_init_device());
  r0h_ctx* ctx = new r0h_ctx();
  ctx->device = device;
  R0H_TRY_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
  for (int d = 0; d < 2; d++) {
    uint32_t w22 = d == 0 ? rou_fwd(TW_TOP) : rou_rev(TW_TOP), w26 = d == 0 ? rou_fwd(MAX_DOMAIN_PO2) : rou_rev(MAX_DOMAIN_PO2);
    R0H_TRY(upload_pow_table(&ctx->twb_lo[d], w26, TWB_SIZE));
    R0H_TRY(upload_pow_table(&ctx->twb_hi[d], fpow(w26, TWB_SIZE), TWB_SIZE));
    R0H_TRY(upload_pow_table(&ctx->tw_lo[d], w22, TW_SIZE));
    R0H_TRY(upload_pow_table(&ctx->tw_hi[d], fpow(w22, TW_SIZE), TW_SIZE));
    {  // the in-chunk table, followed by the same words canonical and by their Shoup companions (ntt.hip, R0H_NTT_SHOUP)
      const uint32_t n12 = 1u << (TWL_BITS - 1), base = d == 0 ? rou_fwd(TWL_BITS) : rou_rev(TWL_BITS);
      std::vector<uint32_t> t(3 * (size_t)n12);
      uint32_t cur = ONE;
      for (uint32_t i = 0; i < n12; i++) { t[i] = cur; t[n12 + i] = dec(cur); t[2 * n12 + i] = shoup_companion(t[n12 + i]); cur = mul(cur, base); }
      R0H_TRY_HIP(hipMalloc((void**)&ctx->tw12[d], t.size() * 4));
      R0H_TRY_HIP(hipMemcpy(ctx->tw12[d], t.data(), t.size() * 4, hipMemcpyHostToDevice));
    }
  }
  R0H_TRY(upload_pow_table(&ctx->pow3_lo, enc(3), TW_SIZE));
  R0H_TRY(upload_pow_table(&ctx->pow3_hi, fpow(enc(3), TW_SIZE), TW_SIZE));
  R0H_TRY(upload_pow_table(&ctx->pow3_top, fpow(enc(3), (uint64_t)1 << TW_TOP), 1u << (MAX_DOMAIN_PO2 - TW_TOP)));
  R0H_TRY_HIP(hipMalloc((void**)&ctx->p2, sizeof(P2Consts)));
  fill_p2(ctx->p2_host, R0H_P2_ROUND_CONSTANTS, R0H_P2_INT_DIAG_M1);
  R0H_TRY_HIP(hipMemcpy(ctx->p2, &ctx->p2_host, sizeof(P2Consts), hipMemcpyHostToDevice));
  ctx->pinned_bytes = 8u << 20;
  R0H_TRY_HIP(hipHostMalloc(&ctx->pinned, ctx->pinned_bytes, hipHostMallocDefault));
  R0H_TRY(ensure_scratch(ctx, 4u << 20));
  *out = ctx;
  return nullptr;
  R0H_GUARD_END
}

const char* r0h_ctx_destroy(r0h_ctx* ctx) {
  if (!ctx) return nullptr;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  ctx_release(ctx);  // buffers and circuits still alive keep the device state until they are freed
  return nullptr;
}

const char* r0h_sync(r0h_ctx* ctx) {
  R0H_REQUIRE(ctx, "r0h_sync: ctx is NULL");
  R0H_TRY_HIP(hipStreamSynchronize(ctx->stream));
  return nullptr;
}

const char* r0h_poseidon2_set_consts(r0h_ctx* ctx, const uint32_t* rc, const uint32_t* diag_m1) {
  R0H_REQUIRE(ctx && rc && diag_m1, "r0h_poseidon2_set_consts: NULL argument");
  for (int i = 0; i < P2_CELLS * P2_ROUNDS; i++) R0H_REQUIRE(rc[i] < P, "round constant %d not canonical", i);
  for (int i = 0; i < P2_CELLS; i++) R0H_REQUIRE(diag_m1[i] < P, "diagonal word %d not canonical", i);
  fill_p2(ctx->p2_host, rc, diag_m1);
  R0H_TRY_HIP(hipStreamSynchronize(ctx->stream));
  R0H_TRY_HIP(hipMemcpy(ctx->p2, &ctx->p2_host, sizeof(P2Consts), hipMemcpyHostToDevice));
  return nullptr;
}

const char* r0h_buf_alloc(r0h_ctx* ctx, size_t bytes, r0h_buf** out) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(ctx && out, "r0h_buf_alloc: NULL argument");
  R0H_TRY_HIP(hipSetDevice(ctx->device));
  r0h_buf* b = new r0h_buf();
  b->ctx = ctx;
  b->bytes = bytes;
  hipError_t e = hipMalloc(&b->ptr, bytes ? bytes : 4);
  if (e != hipSuccess) {
    delete b;
    return make_error("r0h_buf_alloc(%zu bytes): %s", bytes, hipGetErrorString(e));
  }
  ctx_retain(ctx);
  *out = b;
  return nullptr;
  R0H_GUARD_END
}

const char* r0h_buf_wrap(r0h_ctx* ctx, void* device_ptr, size_t bytes, r0h_buf** out) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(ctx && out && device_ptr, "r0h_buf_wrap: NULL argument");
  r0h_buf* b = new r0h_buf();
  b->ctx = ctx; b->ptr = device_ptr; b->bytes = bytes; b->owned = false;
  ctx_retain(ctx);
  *out = b;
  return nullptr;
  R0H_GUARD_END
}

const char* r0h_buf_slice(r0h_buf* parent, size_t off, size_t bytes, r0h_buf** out) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(parent && out, "r0h_buf_slice: NULL argument");
  R0H_REQUIRE(off % 4 == 0 && off + bytes <= parent->bytes, "r0h_buf_slice: [%zu, +%zu) outside a %zu-byte buffer", off, bytes, parent->bytes);
  r0h_buf* b = new r0h_buf();
  b->ctx = parent->ctx; b->ptr = (char*)parent->ptr + off; b->bytes = bytes; b->parent = parent; b->owned = false;
  parent->refs++;
  ctx_retain(parent->ctx);
  *out = b;
  return nullptr;
  R0H_GUARD_END
}

const char* r0h_buf_free(r0h_buf* b) {
  while (b) {
    if (--b->refs > 0) break;
    r0h_buf* parent = b->parent;
    if (b->pooled && b->ptr) {
      // stream-ordered reuse: whoever takes this block next enqueues behind everything that used it
      const size_t sz = b->bytes ? (b->bytes + 255) & ~(size_t)255 : 256;
      if (b->ctx->pool_bytes + sz > POOL_LIMIT) {
        (void)hipStreamSynchronize(b->ctx->stream);
        (void)hipFree(b->ptr);
      } else {
        b->ctx->pool.emplace(sz, b->ptr);
        b->ctx->pool_bytes += sz;
      }
    } else if (b->owned && b->ptr) {
      (void)hipSetDevice(b->ctx->device);  // best effort before the free: a failure here shows up as the free's error
      (void)hipStreamSynchronize(b->ctx->stream);
      hipError_t e = hipFree(b->ptr);
      if (e != hipSuccess) return make_error("r0h_buf_free: %s", hipGetErrorString(e));
    }
    r0h_ctx* ctx = b->ctx;
    delete b;
    ctx_release(ctx);
    b = parent;
  }
  return nullptr;
}

const char* r0h_buf_h2d(r0h_ctx* ctx, r0h_buf* dst, size_t off, const void* src, size_t bytes) {
  R0H_REQUIRE(ctx && dst && (src || !bytes), "r0h_buf_h2d: NULL argument");
  R0H_REQUIRE(off + bytes <= dst->bytes, "r0h_buf_h2d: [%zu, +%zu) outside a %zu-byte buffer", off, bytes, dst->bytes);
  if (!bytes) return nullptr;
  R0H_TRY_HIP(hipMemcpyAsync((char*)dst->ptr + off, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  R0H_TRY_HIP(hipStreamSynchronize(ctx->stream));  // src is pageable caller memory: do not outlive the call
  return nullptr;
}

const char* r0h_buf_d2h(r0h_ctx* ctx, const r0h_buf* src, size_t off, void* dst, size_t bytes) {
  R0H_REQUIRE(ctx && src && (dst || !bytes), "r0h_buf_d2h: NULL argument");
  R0H_REQUIRE(off + bytes <= src->bytes, "r0h_buf_d2h: [%zu, +%zu) outside a %zu-byte buffer", off, bytes, src->bytes);
  if (!bytes) return nullptr;
  R0H_TRY_HIP(hipMemcpyAsync(dst, (const char*)src->ptr + off, bytes, hipMemcpyDeviceToHost, ctx->stream));
  R0H_TRY_HIP(hipStreamSynchronize(ctx->stream));
  return nullptr;
}

const char* r0h_buf_zero(r0h_ctx* ctx, r0h_buf* buf) {
  R0H_REQUIRE(ctx && buf, "r0h_buf_zero: NULL argument");
  R0H_TRY_HIP(hipMemsetAsync(buf->ptr, 0, buf->bytes, ctx->stream));
  return nullptr;
}

const char* r0h_ctx_set_session_resident_limit(r0h_ctx* ctx, uint64_t bytes) {
  R0H_REQUIRE(ctx, "r0h_ctx_set_session_resident_limit: ctx is NULL");
  ctx->session_resident_limit = bytes;
  return nullptr;
}
const char* r0h_kernel_timing(r0h_ctx* ctx, int enable) {
  R0H_REQUIRE(ctx, "r0h_kernel_timing: ctx is NULL");
  R0H_TRY_HIP(hipStreamSynchronize(ctx->stream));
  ctx->ktime_on = enable != 0;
  for (auto& kv : ctx->ktimers) { kv.second.used = 0; kv.second.alg_bytes = 0; }
  return nullptr;
}

const char* r0h_kernel_stats(r0h_ctx* ctx, char* json_out, size_t capacity) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(ctx && json_out && capacity, "r0h_kernel_stats: NULL argument");
  R0H_TRY_HIP(hipStreamSynchronize(ctx->stream));
  std::string js = "{";
  bool first = true;
  for (auto& kv : ctx->ktimers) {
    double total = 0;
    for (size_t i = 0; i + 1 < kv.second.used; i += 2) {
      float ms = 0;
      R0H_TRY_HIP(hipEventElapsedTime(&ms, kv.second.ev[i], kv.second.ev[i + 1]));
      total += ms;
    }
    char item[256];
    snprintf(item, sizeof item, "%s\"%s\": {\"launches\": %zu, \"total_ms\": %.6f, \"alg_bytes\": %.0f}", first ? "" : ", ",
             kv.first.c_str(), kv.second.used / 2, total, kv.second.alg_bytes);
    js += item;
    first = false;
  }
  js += "}";
  R0H_REQUIRE(js.size() + 1 <= capacity, "r0h_kernel_stats: need %zu bytes", js.size() + 1);
  memcpy(json_out, js.c_str(), js.size() + 1);
  return nullptr;
  R0H_GUARD_END
}

void* r0h_buf_device_ptr(const r0h_buf* buf) { return buf ? buf->ptr : nullptr; }
size_t r0h_buf_bytes(const r0h_buf* buf) { return buf ? buf->bytes : 0; }

}  // extern "C"
