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

// One modular add inside a larger asm statement: r = a + b mod p (x: scratch).  r may be a or b.
#define R0H_ASM_ADD(r, a, b, x) "v_add_u32 " r ", " a ", " b "\n\tv_subrev_co_u32 " x ", vcc, 0x78000001, " r "\n\tv_cndmask_b32 " r ", " x ", " r ", vcc\n\t"
// External linear layer: circ(2 M4, M4, .., M4) with M4 = [[2,3,1,1],[1,2,3,1],[1,1,2,3],[3,1,1,2]] -- 128 modular adds.
// Written as seven asm statements (one per block of four cells incl. its share of the column sums, one for the final 24 adds):
// hipcc puts an `s_nop 0` between an asm statement and the next instruction touching a register that statement wrote (it assumes
// a dst-forwarding hazard on gfx950 for anything inside asm), so the one-statement-per-add form carried a nop with every add:
// the kernel is VALU-throughput bound, so the nops themselves cost next to nothing, but the statement form needs 98 instead of 139
// VGPRs (four waves per SIMD instead of three) and measures 0.45 % faster on hash_rows in an A/B on one box
// (profiles/r02/poseidon2_variants.md).  Same canonical words: every add is reduced, only the statement boundaries moved.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(R0H_MEXT_PLAIN)
template <bool FIRST>
__device__ __forceinline__ void m4_block(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t& s0, uint32_t& s1, uint32_t& s2, uint32_t& s3) {
  uint32_t o0, o1, o2, o3, t0, t1, t2, t3, u, x;
  // %0-3 o, %4-7 s (in/out), %8-13 t0 t1 t2 t3 u x, %14-17 a b d e
  asm(R0H_ASM_ADD("%8", "%14", "%15", "%13")   // t0 = a + b
      R0H_ASM_ADD("%9", "%16", "%17", "%13")   // t1 = d + e
      R0H_ASM_ADD("%12", "%15", "%15", "%13")  // u = 2b
      R0H_ASM_ADD("%10", "%12", "%9", "%13")   // t2 = 2b + t1
      R0H_ASM_ADD("%12", "%17", "%17", "%13")  // u = 2e
      R0H_ASM_ADD("%11", "%12", "%8", "%13")   // t3 = 2e + t0
      R0H_ASM_ADD("%9", "%9", "%9", "%13")     // 2 t1
      R0H_ASM_ADD("%9", "%9", "%9", "%13")     // 4 t1
      R0H_ASM_ADD("%3", "%9", "%11", "%13")    // o3 = t4 = 4 t1 + t3
      R0H_ASM_ADD("%8", "%8", "%8", "%13")     // 2 t0
      R0H_ASM_ADD("%8", "%8", "%8", "%13")     // 4 t0
      R0H_ASM_ADD("%1", "%8", "%10", "%13")    // o1 = t5 = 4 t0 + t2
      R0H_ASM_ADD("%0", "%11", "%1", "%13")    // o0 = t3 + t5
      R0H_ASM_ADD("%2", "%10", "%3", "%13")    // o2 = t2 + t4
      R0H_ASM_ADD("%4", "%4", "%0", "%13") R0H_ASM_ADD("%5", "%5", "%1", "%13") R0H_ASM_ADD("%6", "%6", "%2", "%13") R0H_ASM_ADD("%7", "%7", "%3", "%13")
      : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3), "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(u), "=&v"(x)
      : "v"(c0), "v"(c1), "v"(c2), "v"(c3)
      : "vcc");
  c0 = o0; c1 = o1; c2 = o2; c3 = o3;
}
__device__ __forceinline__ void m_ext(uint32_t (&c)[P2_CELLS]) {
  uint32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0, x;
#pragma unroll
  for (int k = 0; k < P2_CELLS; k += 4) m4_block<false>(c[k], c[k + 1], c[k + 2], c[k + 3], s0, s1, s2, s3);
  // %0-23 cells (in/out), %24 scratch, %25-28 column sums
  asm(R0H_ASM_ADD("%0", "%0", "%25", "%24") R0H_ASM_ADD("%1", "%1", "%26", "%24") R0H_ASM_ADD("%2", "%2", "%27", "%24") R0H_ASM_ADD("%3", "%3", "%28", "%24")
      R0H_ASM_ADD("%4", "%4", "%25", "%24") R0H_ASM_ADD("%5", "%5", "%26", "%24") R0H_ASM_ADD("%6", "%6", "%27", "%24") R0H_ASM_ADD("%7", "%7", "%28", "%24")
      R0H_ASM_ADD("%8", "%8", "%25", "%24") R0H_ASM_ADD("%9", "%9", "%26", "%24") R0H_ASM_ADD("%10", "%10", "%27", "%24") R0H_ASM_ADD("%11", "%11", "%28", "%24")
      R0H_ASM_ADD("%12", "%12", "%25", "%24") R0H_ASM_ADD("%13", "%13", "%26", "%24") R0H_ASM_ADD("%14", "%14", "%27", "%24") R0H_ASM_ADD("%15", "%15", "%28", "%24")
      R0H_ASM_ADD("%16", "%16", "%25", "%24") R0H_ASM_ADD("%17", "%17", "%26", "%24") R0H_ASM_ADD("%18", "%18", "%27", "%24") R0H_ASM_ADD("%19", "%19", "%28", "%24")
      R0H_ASM_ADD("%20", "%20", "%25", "%24") R0H_ASM_ADD("%21", "%21", "%26", "%24") R0H_ASM_ADD("%22", "%22", "%27", "%24") R0H_ASM_ADD("%23", "%23", "%28", "%24")
      : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]), "+v"(c[8]), "+v"(c[9]), "+v"(c[10]), "+v"(c[11]),
        "+v"(c[12]), "+v"(c[13]), "+v"(c[14]), "+v"(c[15]), "+v"(c[16]), "+v"(c[17]), "+v"(c[18]), "+v"(c[19]), "+v"(c[20]), "+v"(c[21]), "+v"(c[22]), "+v"(c[23]),
        "=&v"(x)
      : "v"(s0), "v"(s1), "v"(s2), "v"(s3)
      : "vcc");
}
#else
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
#endif

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
