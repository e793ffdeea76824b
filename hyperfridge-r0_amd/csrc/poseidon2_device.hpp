// Device-side Poseidon2 permutation (BabyBear, t=24, x^7, 4+21+4 rounds) over a state held in 24 VGPRs of one lane.
// Shared by poseidon2.hip (hash_rows / hash_fold) and tools/microbench/p2_rounds_bench.hip (per-round cycle counts).
#pragma once
#include "internal.hpp"

namespace r0h {

// x^7 with lazily reduced intermediates (bounds in units of p, x < 1):
//   x2 = x*x        < 0.47 + 1 = 1.47        x3 = x2*x < 0.69 + 1 = 1.69      x4 = x2*x2 < 1.02 + 1 = 2.02 (t + 2^32 p < 2^64 holds)
//   x4' = x4 - p if that does not wrap (< 1.02)                               x7 = x3*x4' < 0.81 + 1, then one reduction
// Two conditional subtractions fewer than four full products; the result is the same canonical word.  (A signed-Montgomery
// chain -- v_mad_i64_i32 / v_mul_hi_i32, no corrections until the end -- measured 19 % slower: tools/microbench/p2_rounds_bench.)
__device__ __forceinline__ uint32_t sbox7(uint32_t x) {
  uint32_t x2 = mul_lazy(x, x);
  uint32_t x3 = mul_lazy(x2, x);
  uint32_t x4 = reduce1(mul_lazy(x2, x2));
  return reduce1(mul_lazy(x3, x4));
}

__device__ __forceinline__ void m_ext(uint32_t (&c)[P2_CELLS]) {
  uint32_t s0, s1, s2, s3;
#pragma unroll
  for (int k = 0; k < P2_CELLS; k += 4) {
    uint32_t a = c[k], b = c[k + 1], d = c[k + 2], e = c[k + 3];
    uint32_t t0 = add(a, b), t1 = add(d, e);
    uint32_t t2 = add(add(b, b), t1), t3 = add(add(e, e), t0);
    uint32_t t1x2 = add(t1, t1), t0x2 = add(t0, t0);
    uint32_t t4 = add(add(t1x2, t1x2), t3), t5 = add(add(t0x2, t0x2), t2);
    c[k] = add(t3, t5); c[k + 1] = t5; c[k + 2] = add(t2, t4); c[k + 3] = t4;
    if (k == 0) { s0 = c[0]; s1 = c[1]; s2 = c[2]; s3 = c[3]; }
    else { s0 = add(s0, c[k]); s1 = add(s1, c[k + 1]); s2 = add(s2, c[k + 2]); s3 = add(s3, c[k + 3]); }
  }
#pragma unroll
  for (int k = 0; k < P2_CELLS; k += 4) {
    c[k] = add(c[k], s0); c[k + 1] = add(c[k + 1], s1); c[k + 2] = add(c[k + 2], s2); c[k + 3] = add(c[k + 3], s3);
  }
}

// c[1] + ... + c[23] mod p for reduced words, as a balanced tree
__device__ __forceinline__ uint32_t sum_lanes_1_to_23(const uint32_t (&c)[P2_CELLS]) {
  uint32_t s1[12];
#pragma unroll
  for (int i = 0; i < 11; i++) s1[i] = add(c[2 * i + 2], c[2 * i + 3]);
  s1[11] = c[1];
  uint32_t s2[6];
#pragma unroll
  for (int i = 0; i < 6; i++) s2[i] = add(s1[2 * i], s1[2 * i + 1]);
  return add(add(add(s2[0], s2[1]), add(s2[2], s2[3])), add(s2[4], s2[5]));
}

__device__ __forceinline__ void p2_full_round(uint32_t (&c)[P2_CELLS], const uint32_t* __restrict__ rc) {
#pragma unroll
  for (int i = 0; i < P2_CELLS; i++) c[i] = sbox7(add(c[i], rc[i]));
  m_ext(c);
}

// Lazily accumulated dot product mod p: 64-bit accumulator, products of reduced words.  The first four products fit as they
// are (4 p^2 < 2^64); from then on the high word is brought below p before every second product, which keeps the total
// below p 2^32 + 2 p^2 < 2^64 and its high word below 2p.  ~11 SIMD cycles per term against ~34 for a reduced product plus a
// modular add (tools/microbench/dot_bench.hip).  `idx` is the term's position: compile-time after unrolling.
__device__ __forceinline__ void dot_fix(uint64_t& acc) {
  acc = ((uint64_t)reduce1((uint32_t)(acc >> 32)) << 32) | (uint32_t)acc;
  // opaque re-pack: seen as (hi << 32) + lo, the optimiser re-associates the following multiply-adds into separate 64-bit
  // additions and moves instead of one v_mad_u64_u32 on the accumulator pair
  asm("" : "+v"(acc));
}
__device__ __forceinline__ void dot_mac(uint64_t& acc, uint32_t a, uint32_t b, int idx) {
  if (idx == 0) { acc = (uint64_t)a * b; return; }
  if (idx >= 4 && (idx & 1) == 0) dot_fix(acc);
  acc += (uint64_t)a * b;
}
__device__ __forceinline__ uint32_t dot_finish(uint64_t acc) {
  dot_fix(acc);  // below p 2^32 + 2^32 < 4 p^2: what reduce64 asks for
  return reduce64(acc);
}

// The 21 partial rounds with the linear layer of lanes 1..23 deferred.  Only lane 0 is non-linear, so with u_i the lanes at
// entry, D_i = mu_i - 1 and sum_r the state sum of round r, lane i after round r is  D_i^r u_i + sum_{j<r} D_i^{r-1-j} sum_j.
// Per round that leaves the S-box of lane 0 and ONE dot product for the sum of the other lanes,
//   sigma_r = sum_i D_i^r u_i + sum_{j<r} kappa_{r-1-j} sum_j      (kappa_k = sum_i D_i^k),
// and the lanes themselves are materialised once at the end (22 terms each): 1176 lazily accumulated terms in all, instead
// of 21 x 24 reduced constant products and 21 x 48 modular adds.  Same canonical words as the round-by-round form.
__device__ __forceinline__ void p2_partial_rounds(uint32_t (&c)[P2_CELLS], const P2Consts* __restrict__ k) {
  // table offsets go through an opaque zero: with constant offsets, loop-invariant code motion precomputes one 64-bit address
  // per table row outside the caller's loop and spills hundreds of SGPRs
  uint32_t zero;
  asm volatile("s_mov_b32 %0, 0" : "=s"(zero));
  uint32_t sums[P2_PARTIAL];
  uint32_t s0 = c[0];
  uint32_t sigma = sum_lanes_1_to_23(c);
  const uint32_t* __restrict__ row = k->part_sigma + zero;
#pragma unroll
  for (int r = 0; r < P2_PARTIAL; r++) {
    if (r > 0) {
      uint64_t acc;
#pragma unroll
      for (int i = 0; i < P2_CELLS - 1; i++) dot_mac(acc, c[i + 1], row[i], i);
#pragma unroll
      for (int j = 0; j < r; j++) dot_mac(acc, sums[j], row[P2_CELLS - 1 + j], P2_CELLS - 1 + j);
      sigma = dot_finish(acc);
      row += P2_CELLS - 1 + r;
    }
    const uint32_t y = sbox7(add(s0, k->rc_partial[r]));
    sums[r] = add(y, sigma);
    s0 = add(sums[r], mul_const(y, k->diag_canon[0], k->diag_shoup[0]));
  }
  c[0] = s0;
#pragma unroll
  for (int i = 1; i < P2_CELLS; i++) {
    const uint32_t* __restrict__ f = k->part_final + zero + (i - 1) * (P2_PARTIAL + 1);
    uint64_t acc;
    dot_mac(acc, c[i], f[0], 0);
#pragma unroll
    for (int q = 0; q < P2_PARTIAL; q++) dot_mac(acc, sums[q], f[1 + q], 1 + q);
    c[i] = dot_finish(acc);
  }
}

__device__ __forceinline__ void p2_mix(uint32_t (&c)[P2_CELLS], const P2Consts* __restrict__ k) {
  m_ext(c);
#pragma unroll 1
  for (int r = 0; r < P2_HALF_FULL; r++) p2_full_round(c, k->rc_full[r]);
  p2_partial_rounds(c, k);
#pragma unroll 1
  for (int r = P2_HALF_FULL; r < 2 * P2_HALF_FULL; r++) p2_full_round(c, k->rc_full[r]);
}

}  // namespace r0h
