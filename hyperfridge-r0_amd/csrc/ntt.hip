// Batched BabyBear NTTs for gfx950: LDS-staged radix-2 butterflies, two HBM passes for sizes above 2^12.
// Replaces risc0-zkp 3.0.4 hal `batch_interpolate_ntt` / `batch_expand_into_evaluate_ntt` / `batch_bit_reverse` /
// `zk_shift` (core/ntt.rs; CUDA side in risc0-sys 1.5.0) -- SURVEY.md 8(a) a2-a5.
//
// Decomposition (N = 2^n = N1 * N2, N2 = 2^L contiguous, N1 = 2^H strided):
//   forward  (bit-reversed coeffs -> natural evals, DIT):  local size-2^L DITs on contiguous chunks, then for every
//            residue lo a size-2^H DIT over the chunk index with the inter-pass twiddle w_N^(brev_H(hi) * lo) on load;
//   inverse  (natural evals -> bit-reversed coeffs, DIF):  the transpose: strided DIF first (twiddle on store), then
//            local DIFs with the 1/N normalisation folded into the final store.
// The strided pass stages a [2^H][T] tile (T consecutive residues, 64-byte rows at T=16) so every global access is a
// run of T words; the local pass moves whole contiguous chunks.  Column-major batches map to blockIdx.y.
#include <algorithm>

#include "internal.hpp"

namespace r0h {

struct TwTables {
  const uint32_t* lo;    // w22^i            (w26^i for the BIG kernels)
  const uint32_t* hi;    // w22^(i << 11)    (w26^(i << 13))
  const uint32_t* tw12;  // ROU[TWL_BITS]^i, i < 2^(TWL_BITS - 1): the twiddles of every layer inside a contiguous chunk
};

// w_n^e from a two-level table: powers of ROU[22] (2 x 2^11 words) for n <= 22; BIG: powers of ROU[26] (2 x 2^13 words) for
// the outer pass of larger transforms (the host hands over the matching table pair)
template <int BIG = 0>
__device__ __forceinline__ uint32_t omega_n(const TwTables& t, uint32_t e, uint32_t n) {
  if (BIG) {
    const uint32_t E = e << (MAX_DOMAIN_PO2 - n);
    return mul(t.lo[E & (TWB_SIZE - 1)], t.hi[E >> TWB_BITS]);
  }
  const uint32_t E = e << (TW_TOP - n);
  return mul(t.lo[E & (TW_SIZE - 1)], t.hi[E >> TW_BITS]);
}

// One contiguous chunk of 2^L words per block.  DIR 0: DIT layers (expand_bits, L]; DIR 1: DIF layers L..1.
template <int DIR>
__global__ __launch_bounds__(256) void ntt_local_kernel(uint32_t* out, const uint32_t* in /* may alias out */,
                                                         uint32_t L, uint32_t n_out, uint32_t expand_bits,
                                                         const uint32_t* __restrict__ tw12, uint32_t scale) {
  extern __shared__ uint32_t s[];
  const uint32_t size = 1u << L, tid = threadIdx.x;
  const size_t col = blockIdx.y;
  const uint32_t* src = in + (col << (n_out - expand_bits));
  uint32_t* dst = out + (col << n_out);
  const uint32_t base = blockIdx.x << L;
  for (uint32_t i = tid; i < size; i += 256) s[i] = src[(base + i) >> expand_bits];
  __syncthreads();
  if (DIR == 0) {
    for (uint32_t l = expand_bits + 1; l <= L; l++) {
      const uint32_t half = 1u << (l - 1);
      for (uint32_t b = tid; b < size / 2; b += 256) {
        uint32_t j = b & (half - 1), i0 = ((b >> (l - 1)) << l) + j;
        uint32_t a = s[i0], t = mul(s[i0 + half], tw12[j << (TWL_BITS - l)]);
        s[i0] = add(a, t);
        s[i0 + half] = sub(a, t);
      }
      __syncthreads();
    }
  } else {
    for (uint32_t l = L; l >= 1; l--) {
      const uint32_t half = 1u << (l - 1);
      for (uint32_t b = tid; b < size / 2; b += 256) {
        uint32_t j = b & (half - 1), i0 = ((b >> (l - 1)) << l) + j;
        uint32_t a = s[i0], t = s[i0 + half];
        s[i0] = add(a, t);
        s[i0 + half] = mul(sub(a, t), tw12[j << (TWL_BITS - l)]);
      }
      __syncthreads();
    }
  }
  if (DIR == 1) {
    for (uint32_t i = tid; i < size; i += 256) dst[base + i] = mul(s[i], scale);
  } else {
    for (uint32_t i = tid; i < size; i += 256) dst[base + i] = s[i];
  }
}

// A [2^H][T] tile per block: T = 2^tlog consecutive residues lo, every chunk index hi.
template <int DIR>
__global__ __launch_bounds__(512) void ntt_strided_kernel(uint32_t* io, uint32_t n, uint32_t L, uint32_t H,
                                                           uint32_t tlog, TwTables tw) {
  extern __shared__ uint32_t s[];
  const uint32_t T = 1u << tlog, tid = threadIdx.x, total = T << H;
  uint32_t* col = io + ((size_t)blockIdx.y << n);
  const uint32_t lo0 = blockIdx.x << tlog;
  for (uint32_t idx = tid; idx < total; idx += 512) {
    uint32_t hi = idx >> tlog, lo = lo0 + (idx & (T - 1));
    uint32_t v = col[((size_t)hi << L) + lo];
    if (DIR == 0) v = mul(v, omega_n(tw, bitrev(hi, H) * lo, n));
    s[idx] = v;
  }
  __syncthreads();
  if (DIR == 0) {
    for (uint32_t l = 1; l <= H; l++) {
      const uint32_t half = 1u << (l - 1);
      for (uint32_t b = tid; b < total / 2; b += 512) {
        uint32_t lo = b & (T - 1), jj = b >> tlog, j = jj & (half - 1);
        uint32_t i0 = ((((jj >> (l - 1)) << l) + j) << tlog) + lo, i1 = i0 + (half << tlog);
        uint32_t a = s[i0], t = mul(s[i1], tw.tw12[j << (TWL_BITS - l)]);
        s[i0] = add(a, t);
        s[i1] = sub(a, t);
      }
      __syncthreads();
    }
  } else {
    for (uint32_t l = H; l >= 1; l--) {
      const uint32_t half = 1u << (l - 1);
      for (uint32_t b = tid; b < total / 2; b += 512) {
        uint32_t lo = b & (T - 1), jj = b >> tlog, j = jj & (half - 1);
        uint32_t i0 = ((((jj >> (l - 1)) << l) + j) << tlog) + lo, i1 = i0 + (half << tlog);
        uint32_t a = s[i0], t = s[i1];
        s[i0] = add(a, t);
        s[i1] = mul(sub(a, t), tw.tw12[j << (TWL_BITS - l)]);
      }
      __syncthreads();
    }
  }
  for (uint32_t idx = tid; idx < total; idx += 512) {
    uint32_t hi = idx >> tlog, lo = lo0 + (idx & (T - 1));
    uint32_t v = s[idx];
    if (DIR == 1) v = mul(v, omega_n(tw, bitrev(hi, H) * lo, n));
    col[((size_t)hi << L) + lo] = v;
  }
}


// ------------------------------------------------------------------ radix-16 register-blocked kernels
// Sub-transforms of size 2^m (8 <= m <= 12) are done in rounds of four butterfly layers: every thread keeps 16 words
// in VGPRs, runs the layers of one 4-bit index field on them, and the block exchanges through LDS between rounds
// (2 LDS writes + 2 LDS reads per word per pass instead of 2 per layer).  Twiddles of a round are
// ROU[l]^(low bits) -- one table word per layer and thread -- times a constant 16th root of unity.
#ifndef R0H_NTT_SHOUP
#define R0H_NTT_SHOUP 0  // A/B: table and constant twiddles in Shoup form (fp.hpp mul_const); see profiles/r04/ntt_shoup_ab.md
#endif
struct W16 {
  uint32_t w[8];  // ROU[4]^k, k < 8 (forward or inverse)
#if R0H_NTT_SHOUP
  uint32_t wc[8], ws[8];  // the same constants canonical, and their Shoup companions floor(w 2^32 / p)
#endif
};
#if R0H_NTT_SHOUP
constexpr uint32_t TW12_WORDS = 1u << (TWL_BITS - 1);  // tw12 is followed by its canonical form and its Shoup companions (ctx.cpp)
#endif

// Layers of the field at bit B (width W) on 16 >> W independent sets; DIT when DIR == 0 (skipping layers below
// LO_LAYER), DIF when DIR == 1.  rest0 = index of the thread's first set among the 2^(m-W) sets of the sub-transform.
template <int W, int B, int DIR, int LO_LAYER>
__device__ __forceinline__ void field_layers(uint32_t (&x)[16], uint32_t rest0, const uint32_t* __restrict__ tw12, const W16& c) {
  constexpr int SETS = 16 >> W;
#pragma unroll
  for (int s = 0; s < SETS; s++) {
    const uint32_t low = (rest0 + s) & ((1u << B) - 1u);
#pragma unroll
    for (int step = 0; step < W; step++) {
      const int l = DIR == 0 ? step : W - 1 - step;
      if (DIR == 0 && l < LO_LAYER) continue;
      uint32_t a_tw = ONE;
      if (B != 0) a_tw = tw12[low << (TWL_BITS - (B + l + 1))];
#if R0H_NTT_SHOUP
      uint32_t a_c = 1u, a_s = 0u;  // the table word canonical, and its Shoup companion
      if (B != 0) { a_c = tw12[TW12_WORDS + (low << (TWL_BITS - (B + l + 1)))]; a_s = tw12[2 * TW12_WORDS + (low << (TWL_BITS - (B + l + 1)))]; }
      auto times = [&](uint32_t y, int cidx) {  // y times the butterfly's twiddle: one Shoup product per factor that is not one
        if (B != 0) y = mul_const(y, a_c, a_s);
        if (cidx != 0) y = mul_const(y, c.wc[cidx], c.ws[cidx]);
        return y;
      };
#endif
#pragma unroll
      for (int j0 = 0; j0 < (1 << W); j0++) {
        if (j0 & (1 << l)) continue;
        const int j1 = j0 | (1 << l), cidx = (j0 & ((1 << l) - 1)) << (3 - l);
        uint32_t& u = x[s * (1 << W) + j0];
        uint32_t& v = x[s * (1 << W) + j1];
        const bool trivial = B == 0 && cidx == 0;
#if R0H_NTT_SHOUP
        if (DIR == 0) {
          uint32_t a = u, t = trivial ? v : times(v, cidx);
          u = add(a, t);
          v = sub(a, t);
        } else {
          uint32_t a = u, t = v;
          u = add(a, t);
          v = trivial ? sub(a, t) : times(sub(a, t), cidx);
        }
#else
        uint32_t twd = cidx == 0 ? a_tw : (B == 0 ? c.w[cidx] : mul_lazy(a_tw, c.w[cidx]));  // < 2p is fine as a multiplier
        if (DIR == 0) {
          uint32_t a = u, t = trivial ? v : mul(v, twd);
          u = add(a, t);
          v = sub(a, t);
        } else {
          uint32_t a = u, t = v;
          u = add(a, t);
          v = trivial ? sub(a, t) : mul(sub(a, t), twd);
        }
#endif
      }
    }
  }
}

// element index of word (set s, digit j) for the field at bit B of width W, thread's first set rest0
template <int W, int B>
__device__ __forceinline__ uint32_t field_index(uint32_t rest0, int s, int j) {
  uint32_t r = rest0 + s;
  return ((r >> B) << (B + W)) | ((uint32_t)j << B) | (r & ((1u << B) - 1u));
}

// Inter-pass twiddles of one thread: element e = q*16 + j needs w_n^(brev_H(e) * lo) with
// brev_H(e) = brev_4(j) << (H-4) | brev_{H-4}(q), i.e. base * g^brev_4(j) with base = w_n^(brev(q) * lo), g = w_n^(lo << (H-4)):
// four table words and 29 products per 16 elements instead of 32 table words and 16 products.
template <uint32_t H, int BIG>
__device__ __forceinline__ void interpass_twiddles(uint32_t (&t)[16], const TwTables& tw, uint32_t q, uint32_t lo, uint32_t n) {
  const uint32_t base = omega_n<BIG>(tw, bitrev(q, H - 4) * lo, n), g = omega_n<BIG>(tw, lo << (H - 4), n);
  uint32_t gp[16];  // base * g^k
  gp[0] = base;
#pragma unroll
  for (int k = 1; k < 16; k++) gp[k] = mul_lazy(gp[k - 1], g);  // multipliers may stay below 2p (g < p keeps the bound)
#pragma unroll
  for (int j = 0; j < 16; j++) t[j] = gp[((j & 1) << 3) | ((j & 2) << 1) | ((j & 4) >> 1) | ((j & 8) >> 3)];
}

// [2^H][T] tile, H = 8 + WL, T = 16 * WORDS consecutive residues: forward = DIT over the chunk index with the inter-pass
// twiddle on load, inverse = DIF with the twiddle on store.  Block = 16 << (H - 4) threads; a thread owns WORDS adjacent
// residues (one radix-16 column each).  WORDS = 2 makes every global access a 128-byte row: 4.8 TB/s against 3.2 TB/s for
// 64-byte rows in a plain copy of the same shape (tools/microbench/tile_copy_bench.hip).  The tile has to stay small enough
// for two blocks per CU (load, butterflies and store of different blocks overlap): H <= 9, i.e. the contiguous pass takes
// up to 2^13 words.  LDS rows are padded by two words so that the four row groups of a wave fall into different banks.
// BIG: the rows are 2^L <= 2^18 words apart and the transform has more than 2^22 points (third level, see the host side).
template <int WL, int DIR, int WORDS, int BIG = 0>
__global__ __launch_bounds__(16 << (4 + WL)) void ntt_strided16_kernel(uint32_t* io, const uint32_t* in, uint32_t n, uint32_t L, TwTables tw, W16 c) {
  extern __shared__ uint32_t s[];
  constexpr uint32_t H = 8 + WL;
  constexpr uint32_t S = WORDS == 1 ? 16 : 16 * WORDS + 2;  // LDS row stride in words
  const uint32_t t = threadIdx.x & 15, q = threadIdx.x >> 4, lo = ((blockIdx.x << 4) + t) * WORDS;
  uint32_t* col = io + ((size_t)blockIdx.y << n);
  const uint32_t* src = in + ((size_t)blockIdx.y << n);  // may alias col (in-place): every word is read before its tile is written
  uint32_t x[WORDS][16];
  auto gload = [&](const uint32_t* base, uint32_t row, int j) {
    const uint32_t* p = base + ((size_t)row << L) + lo;
    if (WORDS == 1) x[0][j] = p[0];
    else { const uint2 v = *(const uint2*)p; x[0][j] = v.x; x[WORDS - 1][j] = v.y; }
  };
  auto gstore = [&](uint32_t row, int j) {
    uint32_t* p = col + ((size_t)row << L) + lo;
    if (WORDS == 1) p[0] = x[0][j];
    else *(uint2*)p = make_uint2(x[0][j], x[WORDS - 1][j]);
  };
  auto sput = [&](uint32_t row, int j) {
    uint32_t* p = s + row * S + t * WORDS;
    if (WORDS == 1) p[0] = x[0][j];
    else *(uint2*)p = make_uint2(x[0][j], x[WORDS - 1][j]);
  };
  auto sget = [&](uint32_t row, int j) {
    const uint32_t* p = s + row * S + t * WORDS;
    if (WORDS == 1) x[0][j] = p[0];
    else { const uint2 v = *(const uint2*)p; x[0][j] = v.x; x[WORDS - 1][j] = v.y; }
  };
  constexpr int WLs = WL == 0 ? 1 : WL, SETS = 16 >> WLs;
  if (DIR == 0) {
#pragma unroll
    for (int j = 0; j < 16; j++) gload(src, q * 16 + j, j);
#pragma unroll
    for (int w = 0; w < WORDS; w++) {
      uint32_t tws[16];
      interpass_twiddles<H, BIG>(tws, tw, q, lo + w, n);
#pragma unroll
      for (int j = 0; j < 16; j++) x[w][j] = mul(x[w][j], tws[j]);
      field_layers<4, 0, 0, 0>(x[w], q, tw.tw12, c);
    }
#pragma unroll
    for (int j = 0; j < 16; j++) sput(q * 16 + j, j);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; j++) sget(field_index<4, 4>(q, 0, j), j);
#pragma unroll
    for (int w = 0; w < WORDS; w++) field_layers<4, 4, 0, 0>(x[w], q, tw.tw12, c);
    if (WL == 0) {
#pragma unroll
      for (int j = 0; j < 16; j++) gstore(field_index<4, 4>(q, 0, j), j);
      return;
    }
#pragma unroll
    for (int j = 0; j < 16; j++) sput(field_index<4, 4>(q, 0, j), j);
    __syncthreads();
#pragma unroll
    for (int ss = 0; ss < SETS; ss++)
#pragma unroll
      for (int j = 0; j < (1 << WLs); j++) sget(field_index<WLs, 8>(q * SETS, ss, j), ss * (1 << WLs) + j);
#pragma unroll
    for (int w = 0; w < WORDS; w++) field_layers<WLs, 8, 0, 0>(x[w], q * SETS, tw.tw12, c);
#pragma unroll
    for (int ss = 0; ss < SETS; ss++)
#pragma unroll
      for (int j = 0; j < (1 << WLs); j++) gstore(field_index<WLs, 8>(q * SETS, ss, j), ss * (1 << WLs) + j);
  } else {
    if (WL != 0) {
#pragma unroll
      for (int ss = 0; ss < SETS; ss++)
#pragma unroll
        for (int j = 0; j < (1 << WLs); j++) gload(src, field_index<WLs, 8>(q * SETS, ss, j), ss * (1 << WLs) + j);
#pragma unroll
      for (int w = 0; w < WORDS; w++) field_layers<WLs, 8, 1, 0>(x[w], q * SETS, tw.tw12, c);
#pragma unroll
      for (int ss = 0; ss < SETS; ss++)
#pragma unroll
        for (int j = 0; j < (1 << WLs); j++) sput(field_index<WLs, 8>(q * SETS, ss, j), ss * (1 << WLs) + j);
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 16; j++) sget(field_index<4, 4>(q, 0, j), j);
    } else {
#pragma unroll
      for (int j = 0; j < 16; j++) gload(src, field_index<4, 4>(q, 0, j), j);
    }
#pragma unroll
    for (int w = 0; w < WORDS; w++) field_layers<4, 4, 1, 0>(x[w], q, tw.tw12, c);
#pragma unroll
    for (int j = 0; j < 16; j++) sput(field_index<4, 4>(q, 0, j), j);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; j++) sget(q * 16 + j, j);
#pragma unroll
    for (int w = 0; w < WORDS; w++) {
      field_layers<4, 0, 1, 0>(x[w], q, tw.tw12, c);
      uint32_t tws[16];
      interpass_twiddles<H, BIG>(tws, tw, q, lo + w, n);
#pragma unroll
      for (int j = 0; j < 16; j++) x[w][j] = mul(x[w][j], tws[j]);
    }
#pragma unroll
    for (int j = 0; j < 16; j++) gstore(q * 16 + j, j);
  }
}

// Third level of transforms above 2^23 points: the top four bits of the index -- 16 rows, 2^(n-4) words apart.  A thread owns one
// residue: sixteen words, the inter-pass twiddles w_n^(brev_4(j) * lo) as powers of one table product, one radix-16 butterfly in
// registers.  No LDS; every access of a wave is a 256-byte run per row, sixteen of them in flight per lane.
template <int DIR>
__global__ __launch_bounds__(256) void ntt_outer16_kernel(uint32_t* io, const uint32_t* in /* may alias io */, uint32_t n, TwTables twb, W16 c) {
  const uint32_t L = n - 4, lo = blockIdx.x * 256 + threadIdx.x;
  uint32_t* col = io + ((size_t)blockIdx.y << n) + lo;
  const uint32_t* src = in + ((size_t)blockIdx.y << n) + lo;
  uint32_t x[16], gp[16];
#pragma unroll
  for (int j = 0; j < 16; j++) x[j] = src[(size_t)j << L];
  gp[0] = ONE;
  gp[1] = omega_n<1>(twb, lo, n);
#pragma unroll
  for (int k = 2; k < 16; k++) gp[k] = mul_lazy(gp[k - 1], gp[1]);  // multipliers may stay below 2p
  if (DIR == 1) field_layers<4, 0, 1, 0>(x, 0, twb.tw12, c);
#pragma unroll
  for (int j = 1; j < 16; j++) x[j] = mul(x[j], gp[((j & 1) << 3) | ((j & 2) << 1) | ((j & 4) >> 1) | ((j & 8) >> 3)]);
  if (DIR == 0) field_layers<4, 0, 0, 0>(x, 0, twb.tw12, c);
#pragma unroll
  for (int j = 0; j < 16; j++) col[(size_t)j << L] = x[j];
}

__device__ __forceinline__ uint32_t lds_pad(uint32_t e) { return e + (e >> 4); }

// One contiguous chunk of 2^L words (L = 8 + WL) per block of 2^(L-4) threads.
// Forward (DIR 0): DIT, optionally fed by an input EXP_BITS (0 or 2) times shorter (each word replicated 2^EXP_BITS times,
// the first EXP_BITS layers skipped).  Inverse (DIR 1): DIF, result scaled by `scale`.
struct ZkShift {        // optional fused f(x) -> f(3x) on the inverse transform's output
  const uint32_t* lo;   // 3^i, i < 2^11
  const uint32_t* hi;   // 3^(i << 11)
  const uint32_t* top;  // 3^(i << 22), i < 16
  uint32_t outer;       // the columns are the 2^outer blocks of a larger transform (third level): block b of a column holds the
                        // coefficients whose index ends in brev(b)
  uint32_t g[16];       // 3^(k << (n - 4)), k < 16, n = log2 of the whole transform
};

template <int WL, int DIR, int EXP_BITS, int ZK>
__global__ __launch_bounds__(1 << (4 + (WL < 4 ? 4 : WL))) void ntt_local16_kernel(uint32_t* out, const uint32_t* in /* may alias out */, uint32_t n_out,
                                                           const uint32_t* __restrict__ tw12, W16 c, uint32_t scale, ZkShift zk) {
  extern __shared__ uint32_t s[];
  constexpr uint32_t L = 8 + WL;
  // rounds: bits 0-3, 4-7, then WLs more bits at 8; L = 13 (WL = 5) adds a one-layer round at bit 12
  constexpr int WLs = WL == 0 ? 1 : (WL > 4 ? 4 : WL), SETS = 16 >> WLs;
  const uint32_t q = threadIdx.x;
  const size_t colid = blockIdx.y;
  const uint32_t base = blockIdx.x << L;
  uint32_t* dst = out + (colid << n_out) + base;
  uint32_t x[16];
  if (DIR == 0) {
    const uint32_t* src = in + (colid << (n_out - EXP_BITS)) + (base >> EXP_BITS);
    if (EXP_BITS == 2) {
      uint4 v = *(const uint4*)(src + q * 4);
      const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 16; j++) x[j] = w4[j >> 2];
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        uint4 v = *(const uint4*)(src + q * 16 + 4 * k);
        x[4 * k] = v.x; x[4 * k + 1] = v.y; x[4 * k + 2] = v.z; x[4 * k + 3] = v.w;
      }
    }
    field_layers<4, 0, 0, EXP_BITS>(x, q, tw12, c);
#pragma unroll
    for (int j = 0; j < 16; j++) s[lds_pad(q * 16 + j)] = x[j];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; j++) x[j] = s[lds_pad(field_index<4, 4>(q, 0, j))];
    field_layers<4, 4, 0, 0>(x, q, tw12, c);
    if (WL == 0) {
#pragma unroll
      for (int j = 0; j < 16; j++) dst[field_index<4, 4>(q, 0, j)] = x[j];
      return;
    }
#pragma unroll
    for (int j = 0; j < 16; j++) s[lds_pad(field_index<4, 4>(q, 0, j))] = x[j];
    __syncthreads();
#pragma unroll
    for (int ss = 0; ss < SETS; ss++)
#pragma unroll
      for (int j = 0; j < (1 << WLs); j++) x[ss * (1 << WLs) + j] = s[lds_pad(field_index<WLs, 8>(q * SETS, ss, j))];
    field_layers<WLs, 8, 0, 0>(x, q * SETS, tw12, c);
    if (WL == 5) {
#pragma unroll
      for (int j = 0; j < 16; j++) s[lds_pad(field_index<4, 8>(q, 0, j))] = x[j];
      __syncthreads();
#pragma unroll
      for (int ss = 0; ss < 8; ss++)
#pragma unroll
        for (int j = 0; j < 2; j++) x[ss * 2 + j] = s[lds_pad(field_index<1, 12>(q * 8, ss, j))];
      field_layers<1, 12, 0, 0>(x, q * 8, tw12, c);
#pragma unroll
      for (int ss = 0; ss < 8; ss++)
#pragma unroll
        for (int j = 0; j < 2; j++) dst[field_index<1, 12>(q * 8, ss, j)] = x[ss * 2 + j];
      return;
    }
#pragma unroll
    for (int ss = 0; ss < SETS; ss++)
#pragma unroll
      for (int j = 0; j < (1 << WLs); j++) dst[field_index<WLs, 8>(q * SETS, ss, j)] = x[ss * (1 << WLs) + j];
  } else {
    const uint32_t* src = in + (colid << n_out) + base;
    if (WL != 0) {
      if (WL == 5) {  // the one-layer round at bit 12 first, then through LDS into the bit-8 layout
#pragma unroll
        for (int ss = 0; ss < 8; ss++)
#pragma unroll
          for (int j = 0; j < 2; j++) x[ss * 2 + j] = src[field_index<1, 12>(q * 8, ss, j)];
        field_layers<1, 12, 1, 0>(x, q * 8, tw12, c);
#pragma unroll
        for (int ss = 0; ss < 8; ss++)
#pragma unroll
          for (int j = 0; j < 2; j++) s[lds_pad(field_index<1, 12>(q * 8, ss, j))] = x[ss * 2 + j];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; j++) x[j] = s[lds_pad(field_index<4, 8>(q, 0, j))];
        __syncthreads();  // the next exchange reuses the buffer
      } else {
#pragma unroll
        for (int ss = 0; ss < SETS; ss++)
#pragma unroll
          for (int j = 0; j < (1 << WLs); j++) x[ss * (1 << WLs) + j] = src[field_index<WLs, 8>(q * SETS, ss, j)];
      }
      field_layers<WLs, 8, 1, 0>(x, q * SETS, tw12, c);
#pragma unroll
      for (int ss = 0; ss < SETS; ss++)
#pragma unroll
        for (int j = 0; j < (1 << WLs); j++) s[lds_pad(field_index<WLs, 8>(q * SETS, ss, j))] = x[ss * (1 << WLs) + j];
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 16; j++) x[j] = s[lds_pad(field_index<4, 4>(q, 0, j))];
    } else {
#pragma unroll
      for (int j = 0; j < 16; j++) x[j] = src[field_index<4, 4>(q, 0, j)];
    }
    field_layers<4, 4, 1, 0>(x, q, tw12, c);
#pragma unroll
    for (int j = 0; j < 16; j++) s[lds_pad(field_index<4, 4>(q, 0, j))] = x[j];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; j++) x[j] = s[lds_pad(q * 16 + j)];
    field_layers<4, 0, 1, 0>(x, q, tw12, c);
    if (ZK) {
      // position p = base + 16 q + j holds the coefficient of x^brev_n(p), brev_n(p) = brev_n(base + 16 q) + (brev_4(j) << (n - 4));
      // as block b of a transform 2^outer times larger: x^(brev_n(p) << outer | brev_outer(b))
      const uint32_t e0 = (bitrev(base + q * 16, n_out) << zk.outer) | bitrev((uint32_t)colid & ((1u << zk.outer) - 1u), zk.outer);
      const uint32_t s0 = mul(scale, mul(mul(zk.lo[e0 & (TW_SIZE - 1)], zk.hi[(e0 >> TW_BITS) & (TW_SIZE - 1)]), zk.top[e0 >> TW_TOP]));
#pragma unroll
      for (int j = 0; j < 16; j++) x[j] = mul(x[j], mul(s0, zk.g[((j & 1) << 3) | ((j & 2) << 1) | ((j & 4) >> 1) | ((j & 8) >> 3)]));
#pragma unroll
      for (int k = 0; k < 4; k++) *(uint4*)(dst + q * 16 + 4 * k) = make_uint4(x[4 * k], x[4 * k + 1], x[4 * k + 2], x[4 * k + 3]);
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++)
        *(uint4*)(dst + q * 16 + 4 * k) = make_uint4(mul(x[4 * k], scale), mul(x[4 * k + 1], scale), mul(x[4 * k + 2], scale), mul(x[4 * k + 3], scale));
    }
  }
}

__global__ void bit_reverse_kernel(uint32_t* io, uint32_t po2) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t* col = io + ((size_t)blockIdx.y << po2);
  uint32_t j = bitrev(i, po2);
  if (i < j) {
    uint32_t a = col[i], b = col[j];
    col[i] = b;
    col[j] = a;
  }
}

// Tiled in-place bit reversal for po2 >= 10: i = (a:5 | m:po2-10 | b:5) maps to (brev b | brev m | brev a), so the 32x32
// tile of middle index m lands, transposed and index-reversed, in the tile of brev(m).  A block swaps the pair
// (m, brev m) through LDS: every global access is a 128-byte row.
__global__ __launch_bounds__(256) void bit_reverse_tiled_kernel(uint32_t* io, uint32_t po2) {
  __shared__ uint32_t ta[32][33], tb[32][33];
  const uint32_t mbits = po2 - 10, m = blockIdx.x, mr = bitrev(m, mbits);
  if (m > mr) return;
  uint32_t* col = io + ((size_t)blockIdx.y << po2);
  const uint32_t b = threadIdx.x & 31, a0 = threadIdx.x >> 5;
#pragma unroll
  for (uint32_t k = 0; k < 4; k++) {
    uint32_t a = a0 + 8 * k;
    ta[a][b] = col[((size_t)a << (po2 - 5)) + (m << 5) + b];
    if (m != mr) tb[a][b] = col[((size_t)a << (po2 - 5)) + (mr << 5) + b];
  }
  __syncthreads();
  const uint32_t rb = __brev(b) >> 27;
#pragma unroll
  for (uint32_t k = 0; k < 4; k++) {
    uint32_t a = a0 + 8 * k, ra = __brev(a) >> 27;
    // new tile(mr)[a][b] = old tile(m)[brev b][brev a]; and symmetrically
    col[((size_t)a << (po2 - 5)) + (mr << 5) + b] = ta[rb][ra];
    if (m != mr) col[((size_t)a << (po2 - 5)) + (m << 5) + b] = tb[rb][ra];
  }
}

// the same pair-of-tiles swap for 16-byte (extension) elements: 16x16 tiles, 256-byte rows
__global__ __launch_bounds__(256) void bit_reverse_ext_tiled_kernel(uint4* io, uint32_t po2) {
  __shared__ uint4 ta[16][17], tb[16][17];
  const uint32_t mbits = po2 - 8, m = blockIdx.x, mr = bitrev(m, mbits);
  if (m > mr) return;
  uint4* col = io + ((size_t)blockIdx.y << po2);
  const uint32_t b = threadIdx.x & 15, a = threadIdx.x >> 4;
  ta[a][b] = col[((size_t)a << (po2 - 4)) + (m << 4) + b];
  if (m != mr) tb[a][b] = col[((size_t)a << (po2 - 4)) + (mr << 4) + b];
  __syncthreads();
  const uint32_t rb = __brev(b) >> 28, ra = __brev(a) >> 28;
  col[((size_t)a << (po2 - 4)) + (mr << 4) + b] = ta[rb][ra];
  if (m != mr) col[((size_t)a << (po2 - 4)) + (m << 4) + b] = tb[rb][ra];
}
__global__ void bit_reverse_ext_kernel(uint4* io, uint32_t po2) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  uint4* col = io + ((size_t)blockIdx.y << po2);
  uint32_t j = bitrev(i, po2);
  if (i < j) {
    uint4 a = col[i], b = col[j];
    col[i] = b;
    col[j] = a;
  }
}

__global__ void zk_shift_kernel(uint32_t* io, uint32_t po2, const uint32_t* pow3_lo, const uint32_t* pow3_hi, const uint32_t* pow3_top) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t* col = io + ((size_t)blockIdx.y << po2);
  uint32_t e = bitrev(i, po2);
  uint32_t s = mul(pow3_lo[e & (TW_SIZE - 1)], pow3_hi[(e >> TW_BITS) & (TW_SIZE - 1)]);
  if (po2 > TW_TOP) s = mul(s, pow3_top[e >> TW_TOP]);
  col[i] = mul(col[i], s);
}

struct Split {
  uint32_t L, H, tlog;
};
static Split split_for(uint32_t n) {
  Split sp;
  if (n <= 12) { sp.L = n; sp.H = 0; sp.tlog = 0; return sp; }
  uint32_t H = n / 2;
  if (H > 10) H = 10;
  if (n - H > 12) H = n - 12;
  sp.H = H;
  sp.L = n - H;
  sp.tlog = 14 - H;
  if (sp.tlog > sp.L) sp.tlog = sp.L;
  return sp;
}


struct Split16 {
  bool use16;      // radix-16 kernels apply
  uint32_t L, H;   // contiguous chunk 2^L, strided 2^H (0 = single pass)
  uint32_t outer;  // third level (domains above 2^23): a column is 2^outer blocks of 2^(L + H) words (0 = none)
};
// 2^16 .. 2^22: two passes (H = 8 or 9: two strided tiles per CU).  2^23: two passes with 2^10-row tiles (139 KB of LDS, one
// block per CU) and the ROU[26] tables.  2^24 .. 2^26 (segments of 2^22 .. 2^24 rows -- what 288 GB of HBM has room for): three
// levels.  A column's 16 blocks of 2^20 .. 2^22 contiguous words are transformed like 16 columns by the two tuned passes above,
// and one register-only radix-16 pass runs across the blocks (ntt_outer16_kernel).
static Split16 split16_for(uint32_t n) {
  Split16 sp{false, n, 0, 0};
  if (n >= 8 && n <= 12) { sp.use16 = true; return sp; }
  if (n >= 16 && n <= MAX_DOMAIN_PO2) {
    sp.use16 = true;
    if (n > 23) { sp.outer = 4; n -= 4; }
    sp.H = n <= 20 ? 8 : (n <= 22 ? 9 : 10);  // the contiguous pass takes up to 2^13 words
    sp.L = n - sp.H;
  }
  return sp;
}
static W16 make_w16(bool inverse) {
  W16 c;
  uint32_t w = inverse ? rou_rev(4) : rou_fwd(4), cur = ONE;
  for (int k = 0; k < 8; k++) {
    c.w[k] = cur;
#if R0H_NTT_SHOUP
    c.wc[k] = dec(cur);
    c.ws[k] = shoup_companion(c.wc[k]);
#endif
    cur = mul(cur, w);
  }
  return c;
}
static size_t local16_lds_bytes(uint32_t L) { return (((size_t)1 << L) + ((size_t)1 << (L - 4))) * 4; }

constexpr size_t GRID_Y_MAX = 32768;  // columns per launch (blockIdx.y): the blocks of large transforms count as columns

template <int DIR, int EXP_BITS, int ZK = 0>
static void launch_local16(r0h_ctx* ctx, uint32_t L, uint32_t blocks_x, size_t count, uint32_t* out, const uint32_t* in, uint32_t n_out, const uint32_t* tw12,
                           const W16& c, uint32_t scale, const ZkShift& zk = ZkShift{}) {
  const size_t lds = local16_lds_bytes(L);
  const dim3 block(1u << (L - 4));
  for (size_t c0 = 0; c0 < count; c0 += GRID_Y_MAX) {
    const dim3 grid(blocks_x, (uint32_t)std::min(count - c0, GRID_Y_MAX));
    uint32_t* o = out + (c0 << n_out);
    const uint32_t* i = in + (c0 << (DIR == 0 ? n_out - EXP_BITS : n_out));
    switch (L) {
      case 8: hipLaunchKernelGGL((ntt_local16_kernel<0, DIR, EXP_BITS, ZK>), grid, block, lds, ctx->stream, o, i, n_out, tw12, c, scale, zk); break;
      case 9: hipLaunchKernelGGL((ntt_local16_kernel<1, DIR, EXP_BITS, ZK>), grid, block, lds, ctx->stream, o, i, n_out, tw12, c, scale, zk); break;
      case 10: hipLaunchKernelGGL((ntt_local16_kernel<2, DIR, EXP_BITS, ZK>), grid, block, lds, ctx->stream, o, i, n_out, tw12, c, scale, zk); break;
      case 11: hipLaunchKernelGGL((ntt_local16_kernel<3, DIR, EXP_BITS, ZK>), grid, block, lds, ctx->stream, o, i, n_out, tw12, c, scale, zk); break;
      case 12: hipLaunchKernelGGL((ntt_local16_kernel<4, DIR, EXP_BITS, ZK>), grid, block, lds, ctx->stream, o, i, n_out, tw12, c, scale, zk); break;
      default: hipLaunchKernelGGL((ntt_local16_kernel<5, DIR, EXP_BITS, ZK>), grid, block, lds, ctx->stream, o, i, n_out, tw12, c, scale, zk); break;
    }
  }
}
constexpr size_t strided16_lds_bytes(uint32_t H) { return ((size_t)1 << H) * (16 * 2 + 2) * 4; }
template <int WL, int DIR, int BIG>
static void launch_strided16_wl(r0h_ctx* ctx, uint32_t blocks_x, size_t count, uint32_t* io, const uint32_t* in, uint32_t n, uint32_t L, const TwTables& tw,
                                const W16& c) {
  constexpr uint32_t H = 8 + WL;
  const dim3 block(16u << (H - 4));
  for (size_t c0 = 0; c0 < count; c0 += GRID_Y_MAX) {
    const uint32_t cols = (uint32_t)std::min(count - c0, GRID_Y_MAX);
    uint32_t* o = io + (c0 << n);
    const uint32_t* i = in + (c0 << n);
    if (L >= 5) {  // two residues per thread: 128-byte rows
      // tiles above 64 KB of LDS: the limit was raised for this device when its first context was created (ntt_init_device)
      hipLaunchKernelGGL((ntt_strided16_kernel<WL, DIR, 2, BIG>), dim3(blocks_x / 2, cols), block, strided16_lds_bytes(H), ctx->stream, o, i, n, L, tw, c);
    } else {
      hipLaunchKernelGGL((ntt_strided16_kernel<WL, DIR, 1, BIG>), dim3(blocks_x, cols), block, ((size_t)16 << H) * 4, ctx->stream, o, i, n, L, tw, c);
    }
  }
}
template <int DIR, int BIG>
static void launch_strided16(r0h_ctx* ctx, uint32_t H, uint32_t blocks_x, size_t count, uint32_t* io, const uint32_t* in, uint32_t n, uint32_t L,
                             const TwTables& tw, const W16& c) {
  switch (H) {
    case 8: launch_strided16_wl<0, DIR, BIG>(ctx, blocks_x, count, io, in, n, L, tw, c); break;
    case 9: launch_strided16_wl<1, DIR, BIG>(ctx, blocks_x, count, io, in, n, L, tw, c); break;
    default: launch_strided16_wl<2, DIR, BIG>(ctx, blocks_x, count, io, in, n, L, tw, c); break;
  }
}

template <int DIR>
static void launch_outer16(r0h_ctx* ctx, size_t count, uint32_t* io, const uint32_t* in, uint32_t n, const TwTables& twb, const W16& c) {
  for (size_t c0 = 0; c0 < count; c0 += GRID_Y_MAX) {
    const dim3 grid((1u << (n - 4)) / 256, (uint32_t)std::min(count - c0, GRID_Y_MAX));
    hipLaunchKernelGGL(ntt_outer16_kernel<DIR>, grid, dim3(256), 0, ctx->stream, io + (c0 << n), in + (c0 << n), n, twb, c);
  }
}

// Dynamic LDS above the 64 KB default has to be asked for per kernel instantiation and device.  Done for every instantiation
// that needs it when a context is created (r0h_ctx_create, after hipSetDevice), so no launch can race the request and a
// refusal is reported instead of surfacing later as "launch failed".
template <int WL, int DIR, int BIG>
static const char* raise_lds_limit() {
  constexpr size_t lds = strided16_lds_bytes(8 + WL);
  if (lds <= 65536) return nullptr;
  R0H_TRY_HIP(hipFuncSetAttribute((const void*)ntt_strided16_kernel<WL, DIR, 2, BIG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  return nullptr;
}
const char* ntt_init_device() {
  R0H_TRY((raise_lds_limit<0, 0, 0>()));
  R0H_TRY((raise_lds_limit<0, 1, 0>()));
  R0H_TRY((raise_lds_limit<1, 0, 0>()));
  R0H_TRY((raise_lds_limit<1, 1, 0>()));
  R0H_TRY((raise_lds_limit<2, 0, 0>()));
  R0H_TRY((raise_lds_limit<2, 1, 0>()));
  R0H_TRY((raise_lds_limit<0, 0, 1>()));
  R0H_TRY((raise_lds_limit<0, 1, 1>()));
  R0H_TRY((raise_lds_limit<1, 0, 1>()));
  R0H_TRY((raise_lds_limit<1, 1, 1>()));
  R0H_TRY((raise_lds_limit<2, 0, 1>()));
  R0H_TRY((raise_lds_limit<2, 1, 1>()));
  return nullptr;
}

static const char* launch_check(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return make_error("%s: launch failed: %s", what, hipGetErrorString(e));
  return nullptr;
}

static const char* zk_shift_cols(r0h_ctx* ctx, uint32_t* io, size_t count, uint32_t po2) {
  const uint32_t threads = po2 >= 8 ? 256 : (1u << po2);
  KScope ks(ctx, "zk_shift_kernel", 8.0 * count * (double)((size_t)1 << po2));
  for (size_t c0 = 0; c0 < count; c0 += GRID_Y_MAX) {
    const dim3 grid((1u << po2) / threads, (uint32_t)std::min(count - c0, GRID_Y_MAX));
    hipLaunchKernelGGL(zk_shift_kernel, grid, dim3(threads), 0, ctx->stream, io + (c0 << po2), po2, ctx->pow3_lo, ctx->pow3_hi, ctx->pow3_top);
  }
  return launch_check("zk_shift_kernel");
}

// Radix-16 forward transform of `count` columns (bit-reversed coefficients, 2^(n - expand_bits) words apart at `in`; natural
// evaluations, 2^n words apart at `out`); split16_for(n).use16 holds and expand_bits is 0 or 2.
static const char* forward16(r0h_ctx* ctx, uint32_t* out, const uint32_t* in, size_t count, uint32_t n, uint32_t expand_bits) {
  const Split16 sp = split16_for(n);
  const W16 c = make_w16(false);
  const TwTables tw{ctx->tw_lo[0], ctx->tw_hi[0], ctx->tw12[0]}, twb{ctx->twb_lo[0], ctx->twb_hi[0], ctx->tw12[0]};
  const uint32_t n_in = n - sp.outer;       // what the first two levels transform
  const size_t blocks = count << sp.outer;  // a column's blocks are contiguous on both sides: they are columns of the inner transform
  const double words = (double)((size_t)1 << n);
  {
    KScope ks(ctx, "ntt_local_kernel", 4.0 * count * (words + (double)((size_t)1 << (n - expand_bits))));
    if (expand_bits == 2) launch_local16<0, 2>(ctx, sp.L, 1u << (n_in - sp.L), blocks, out, in, n_in, tw.tw12, c, 0u);
    else launch_local16<0, 0>(ctx, sp.L, 1u << (n_in - sp.L), blocks, out, in, n_in, tw.tw12, c, 0u);
  }
  R0H_TRY(launch_check("ntt_local16_kernel<fwd>"));
  if (sp.H) {
    KScope ks(ctx, "ntt_strided_kernel", 8.0 * count * words);
    if (n_in > TW_TOP) launch_strided16<0, 1>(ctx, sp.H, 1u << (sp.L - 4), blocks, out, out, n_in, sp.L, twb, c);
    else launch_strided16<0, 0>(ctx, sp.H, 1u << (sp.L - 4), blocks, out, out, n_in, sp.L, tw, c);
    R0H_TRY(launch_check("ntt_strided16_kernel<fwd>"));
  }
  if (sp.outer) {
    KScope ks(ctx, "ntt_outer_kernel", 8.0 * count * words);
    launch_outer16<0>(ctx, count, out, out, n, twb, c);
    R0H_TRY(launch_check("ntt_outer16_kernel<fwd>"));
  }
  return nullptr;
}

// The transpose: natural evaluations at `src` (may be `io`) -> bit-reversed coefficients at `io`, scaled by 1/2^n, f(x) -> f(3x) on request
static const char* inverse16(r0h_ctx* ctx, uint32_t* io, const uint32_t* src, size_t count, uint32_t n, bool zk_shift) {
  const Split16 sp = split16_for(n);
  const W16 c = make_w16(true);
  const TwTables tw{ctx->tw_lo[1], ctx->tw_hi[1], ctx->tw12[1]}, twb{ctx->twb_lo[1], ctx->twb_hi[1], ctx->tw12[1]};
  const uint32_t n_in = n - sp.outer, norm = inv(enc(1u << n));
  const size_t blocks = count << sp.outer;
  const double words = (double)((size_t)1 << n);
  const uint32_t* cur = src;
  if (sp.outer) {
    KScope ks(ctx, "ntt_outer_kernel", 8.0 * count * words);
    launch_outer16<1>(ctx, count, io, cur, n, twb, c);
    R0H_TRY(launch_check("ntt_outer16_kernel<inv>"));
    cur = io;
  }
  if (sp.H) {
    KScope ks(ctx, "ntt_strided_kernel", 8.0 * count * words);
    if (n_in > TW_TOP) launch_strided16<1, 1>(ctx, sp.H, 1u << (sp.L - 4), blocks, io, cur, n_in, sp.L, twb, c);
    else launch_strided16<1, 0>(ctx, sp.H, 1u << (sp.L - 4), blocks, io, cur, n_in, sp.L, tw, c);
    R0H_TRY(launch_check("ntt_strided16_kernel<inv>"));
    cur = io;
  }
  {
    KScope ks(ctx, "ntt_local_kernel", 8.0 * count * words);
    if (zk_shift) {  // fused into the last pass: no separate sweep over the coefficients
      ZkShift zk{ctx->pow3_lo, ctx->pow3_hi, ctx->pow3_top, sp.outer, {0}};
      const uint32_t g = fpow(enc(3), (uint64_t)1 << (n - 4));
      uint32_t pw = ONE;
      for (int k = 0; k < 16; k++) { zk.g[k] = pw; pw = mul(pw, g); }
      launch_local16<1, 0, 1>(ctx, sp.L, 1u << (n_in - sp.L), blocks, io, cur, n_in, tw.tw12, c, norm, zk);
    } else {
      launch_local16<1, 0>(ctx, sp.L, 1u << (n_in - sp.L), blocks, io, cur, n_in, tw.tw12, c, norm);
    }
  }
  return launch_check("ntt_local16_kernel<inv>");
}

}  // namespace r0h

using namespace r0h;

extern "C" {

const char* r0h_batch_interpolate_ntt(r0h_ctx* ctx, r0h_buf* io, uint32_t count, uint32_t po2) { return r0h::interpolate_ntt(ctx, io, io, count, po2, false); }
const char* r0h_batch_interpolate_ntt_zk_shift(r0h_ctx* ctx, r0h_buf* io, uint32_t count, uint32_t po2) { return r0h::interpolate_ntt(ctx, io, io, count, po2, true); }

}  // extern "C"

namespace r0h {
const char* interpolate_ntt(r0h_ctx* ctx, r0h_buf* io, const r0h_buf* src, uint32_t count, uint32_t po2, bool zk_shift) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(ctx && io && src, "r0h_batch_interpolate_ntt: NULL argument");
  R0H_REQUIRE(((size_t)count << po2) * 4 <= src->bytes, "r0h_batch_interpolate_ntt: %u columns of 2^%u exceed the source buffer", count, po2);
  R0H_REQUIRE(po2 >= 1 && po2 <= MAX_DOMAIN_PO2, "r0h_batch_interpolate_ntt: po2 %u outside [1, %u]", po2, MAX_DOMAIN_PO2);
  R0H_REQUIRE(((size_t)count << po2) * 4 <= io->bytes, "r0h_batch_interpolate_ntt: %u columns of 2^%u exceed the buffer", count, po2);
  if (!count) return nullptr;
  if (split16_for(po2).use16) return inverse16(ctx, u32(io), u32(src), count, po2, zk_shift);
  const uint32_t norm = inv(enc(1u << po2));
  TwTables tw{ctx->tw_lo[1], ctx->tw_hi[1], ctx->tw12[1]};
  if (src->ptr != io->ptr) R0H_TRY_HIP(hipMemcpyAsync(io->ptr, src->ptr, ((size_t)count << po2) * 4, hipMemcpyDeviceToDevice, ctx->stream));
  const Split sp = split_for(po2);
  if (sp.H) {
    KScope ks(ctx, "ntt_strided_kernel", 8.0 * count * (double)(1u << po2));
    dim3 grid(1u << (sp.L - sp.tlog), count);
    hipLaunchKernelGGL(ntt_strided_kernel<1>, grid, dim3(512), (size_t)4 << (sp.H + sp.tlog), ctx->stream, u32(io), po2, sp.L, sp.H, sp.tlog, tw);
    R0H_TRY(launch_check("ntt_strided_kernel<inv>"));
  }
  KScope ks(ctx, "ntt_local_kernel", 8.0 * count * (double)(1u << po2));
  dim3 grid(1u << (po2 - sp.L), count);
  hipLaunchKernelGGL(ntt_local_kernel<1>, grid, dim3(256), (size_t)4 << sp.L, ctx->stream, u32(io), u32(io), sp.L, po2, 0u, tw.tw12, norm);
  R0H_TRY(launch_check("ntt_local_kernel<inv>"));
  if (zk_shift) return r0h_zk_shift(ctx, io, count, po2);  // small sizes: separate pass
  return nullptr;
  R0H_GUARD_END
}
}  // namespace r0h

extern "C" {

const char* r0h_batch_expand_into_evaluate_ntt(r0h_ctx* ctx, r0h_buf* out, const r0h_buf* in, uint32_t count,
                                               uint32_t in_po2, uint32_t expand_bits) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(ctx && out && in, "r0h_batch_expand_into_evaluate_ntt: NULL argument");
  const uint32_t n = in_po2 + expand_bits;
  R0H_REQUIRE(n >= 1 && n <= MAX_DOMAIN_PO2, "r0h_batch_expand_into_evaluate_ntt: output po2 %u outside [1, %u]", n, MAX_DOMAIN_PO2);
  R0H_REQUIRE(((size_t)count << in_po2) * 4 <= in->bytes && ((size_t)count << n) * 4 <= out->bytes,
              "r0h_batch_expand_into_evaluate_ntt: %u columns exceed the buffers", count);
  R0H_REQUIRE(out->ptr != in->ptr || expand_bits == 0, "r0h_batch_expand_into_evaluate_ntt: in-place expansion is not supported");
  if (!count) return nullptr;
  if (split16_for(n).use16 && (expand_bits == 0 || expand_bits == 2)) return forward16(ctx, u32(out), u32(in), count, n, expand_bits);
  R0H_REQUIRE(n <= TW_TOP, "r0h_batch_expand_into_evaluate_ntt: expand_bits %u is not supported above 2^%u points (0 or 2 are)", expand_bits, TW_TOP);
  TwTables tw{ctx->tw_lo[0], ctx->tw_hi[0], ctx->tw12[0]};
  const Split sp = split_for(n);
  R0H_REQUIRE(expand_bits < sp.L, "r0h_batch_expand_into_evaluate_ntt: expand_bits %u too large for size 2^%u", expand_bits, n);
  dim3 grid(1u << (n - sp.L), count);
  {
    KScope ks(ctx, "ntt_local_kernel", 4.0 * count * ((double)(1u << n) + (double)(1u << in_po2)));
    hipLaunchKernelGGL(ntt_local_kernel<0>, grid, dim3(256), (size_t)4 << sp.L, ctx->stream, u32(out), u32(in), sp.L, n, expand_bits, tw.tw12, 0u);
  }
  R0H_TRY(launch_check("ntt_local_kernel<fwd>"));
  if (sp.H) {
    KScope ks(ctx, "ntt_strided_kernel", 8.0 * count * (double)(1u << n));
    dim3 grid2(1u << (sp.L - sp.tlog), count);
    hipLaunchKernelGGL(ntt_strided_kernel<0>, grid2, dim3(512), (size_t)4 << (sp.H + sp.tlog), ctx->stream, u32(out), n, sp.L, sp.H, sp.tlog, tw);
    R0H_TRY(launch_check("ntt_strided_kernel<fwd>"));
  }
  return nullptr;
  R0H_GUARD_END
}

const char* r0h_batch_bit_reverse(r0h_ctx* ctx, r0h_buf* io, uint32_t count, uint32_t po2) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(ctx && io, "r0h_batch_bit_reverse: NULL argument");
  R0H_REQUIRE(po2 <= MAX_DOMAIN_PO2 + 2, "r0h_batch_bit_reverse: po2 %u too large", po2);
  R0H_REQUIRE(((size_t)count << po2) * 4 <= io->bytes, "r0h_batch_bit_reverse: %u columns of 2^%u exceed the buffer", count, po2);
  if (!count || po2 == 0) return nullptr;
  if (po2 >= 10) {
    KScope ks(ctx, "bit_reverse_kernel", 8.0 * count * (double)(1u << po2));
    hipLaunchKernelGGL(bit_reverse_tiled_kernel, dim3(1u << (po2 - 10), count), dim3(256), 0, ctx->stream, u32(io), po2);
    return launch_check("bit_reverse_tiled_kernel");
  }
  uint32_t threads = po2 >= 8 ? 256 : (1u << po2);
  KScope ks(ctx, "bit_reverse_kernel", 8.0 * count * (double)(1u << po2));
  dim3 grid((1u << po2) / threads, count);
  hipLaunchKernelGGL(bit_reverse_kernel, grid, dim3(threads), 0, ctx->stream, u32(io), po2);
  return launch_check("bit_reverse_kernel");
  R0H_GUARD_END
}

}  // extern "C"
namespace r0h {
const char* bit_reverse_ext(r0h_ctx* ctx, r0h_buf* io, uint32_t count, uint32_t po2) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(ctx && io && po2 <= MAX_DOMAIN_PO2, "bit_reverse_ext: bad argument");
  R0H_REQUIRE(((size_t)count << po2) * 16 <= io->bytes, "bit_reverse_ext: %u columns of 2^%u exceed the buffer", count, po2);
  if (!count || po2 == 0) return nullptr;
  KScope ks(ctx, "bit_reverse_kernel", 32.0 * count * (double)(1u << po2));
  if (po2 >= 8) {
    hipLaunchKernelGGL(bit_reverse_ext_tiled_kernel, dim3(1u << (po2 - 8), count), dim3(256), 0, ctx->stream, (uint4*)io->ptr, po2);
  } else {
    hipLaunchKernelGGL(bit_reverse_ext_kernel, dim3(1, count), dim3(1u << po2), 0, ctx->stream, (uint4*)io->ptr, po2);
  }
  return launch_check("bit_reverse_ext kernel");
  R0H_GUARD_END
}
}  // namespace r0h
extern "C" {

const char* r0h_zk_shift(r0h_ctx* ctx, r0h_buf* io, uint32_t count, uint32_t po2) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(ctx && io, "r0h_zk_shift: NULL argument");
  R0H_REQUIRE(po2 <= MAX_DOMAIN_PO2, "r0h_zk_shift: po2 %u too large", po2);
  R0H_REQUIRE(((size_t)count << po2) * 4 <= io->bytes, "r0h_zk_shift: %u columns of 2^%u exceed the buffer", count, po2);
  if (!count) return nullptr;
  return zk_shift_cols(ctx, u32(io), count, po2);
  R0H_GUARD_END
}

}  // extern "C"
