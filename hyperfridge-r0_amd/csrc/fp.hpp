// BabyBear (p = 15*2^27 + 1) in Montgomery form and its quartic extension, for host and gfx950 device code.
// Replaces risc0-sys 1.5.0 `fp.h` / `fpext.h` and risc0-zkp 3.0.4 `field/baby_bear.rs`.
// 31-bit modular integer work on the VALU: one v_mad_u64_u32 for a*b, one v_mul_lo_u32 for the Montgomery
// quotient, one v_mad_u64_u32 for the reduction, then a branch-free conditional subtract (v_sub + v_min).
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define R0H_HD __host__ __device__ __forceinline__
#else
#define R0H_HD inline
#endif

namespace r0h {

constexpr uint32_t P = 2013265921u;
constexpr uint32_t NPINV = 0x77ffffffu;  // -p^-1 mod 2^32
constexpr uint32_t R2 = 1172168163u;     // 2^64 mod p
constexpr uint32_t ONE = 268435454u;     // 2^32 mod p
constexpr uint32_t ROU_GEN = 137u;       // primitive 2^27-th root (canonical)
constexpr uint32_t BETA_CANON = 11u;

// Conditional correction by p through the carry flag: on gfx950 v_min_u32 / v_max_u32 issue at half rate (4.4 SIMD cycles
// per wave instruction, like every three-operand VOP3 integer op) while v_sub_co_u32 + v_cndmask_b32 are full rate (2.4 each)
// -- tools/microbench/valu_rate_bench.hip, profiles/r01/valu_rate_microbench.txt.  The host build keeps plain C++.
R0H_HD uint32_t reduce1(uint32_t x) {  // x < 2p  ->  x mod p
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t r;
#if defined(R0H_P_IN_VGPR)
  asm("v_sub_co_u32 %0, vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %1, vcc" : "=&v"(r) : "v"(x), "v"(P) : "vcc");
#else
  asm("v_subrev_co_u32 %0, vcc, 0x78000001, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "=&v"(r) : "v"(x) : "vcc");  // r = x - p; borrow ? x : r
#endif
  return r;
#else
  uint32_t y = x - P;
  return y < x ? y : x;  // min as unsigned: x-P wraps above x when x < P
#endif
}
R0H_HD uint32_t add(uint32_t a, uint32_t b) { return reduce1(a + b); }
R0H_HD uint32_t sub(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t d, e;
  asm("v_sub_co_u32 %0, vcc, %2, %3\n\tv_add_u32 %1, 0x78000001, %0\n\tv_cndmask_b32 %0, %0, %1, vcc"
      : "=&v"(d), "=&v"(e) : "v"(a), "v"(b) : "vcc");  // d = a - b; borrow ? d + p : d
  return d;
#else
  uint32_t d = a - b, e = d + P;  // a >= b: d < p <= e;  a < b: d wrapped above 2^32 - p, e = p - (b - a) < p
  return d < e ? d : e;
#endif
}
R0H_HD uint32_t neg(uint32_t a) { return a ? P - a : 0u; }
R0H_HD uint32_t mul(uint32_t a, uint32_t b) {
  uint64_t t = (uint64_t)a * b;
  uint32_t m = (uint32_t)t * NPINV;
  uint64_t u = t + (uint64_t)m * P;
  return reduce1((uint32_t)(u >> 32));
}
// a - b, plus p when it went negative (operands need not be reduced: a < b + p must hold for the result to be below 2p)
R0H_HD uint32_t sub_lazy(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t d, e;
  asm("v_sub_co_u32 %0, vcc, %2, %3\n\tv_add_u32 %1, 0x78000001, %0\n\tv_cndmask_b32 %0, %0, %1, vcc"
      : "=&v"(d), "=&v"(e) : "v"(a), "v"(b) : "vcc");
  return d;
#else
  uint32_t d = a - b;
  return a < b ? d + P : d;
#endif
}
// Montgomery reduction of a sum of up to four products of reduced words (T < 4 p^2 < 2^64): hi(T) - hi(m p) with
// m = lo(T) p^-1.  hi(T) < 4 p^2 / 2^32 = 1.875 p and hi(m p) < p, so after the sign fix one conditional subtraction is enough.
R0H_HD uint32_t reduce64(uint64_t t) {
  uint32_t m = (uint32_t)t * 0x88000001u;  // p^-1 mod 2^32
  uint32_t q = (uint32_t)(((uint64_t)m * P) >> 32);
  uint32_t h = (uint32_t)(t >> 32);
  return reduce1(sub_lazy(h, q));
}
// Product with a known constant (Shoup): a in Montgomery form times the canonical constant w, given w' = floor(w 2^32 / p).
// Result = a*w mod p, i.e. the same word mul(a, enc(w)) returns, in 9 issue slots instead of 12 (no 64-bit products).
R0H_HD uint32_t mul_const(uint32_t a, uint32_t w, uint32_t w_shoup) {
  uint32_t q = (uint32_t)(((uint64_t)a * w_shoup) >> 32);
  return reduce1(a * w - q * P);  // exact in 32 bits: the true value lies in [0, 2p)
}
inline uint32_t shoup_companion(uint32_t w_canonical) { return (uint32_t)(((uint64_t)w_canonical << 32) / P); }
// Montgomery product without the final conditional subtraction.  For a*b + 2^32 p < 2^64 (e.g. a < 2p, b < p; or both
// below 1.47p) the result is congruent to a*b/2^32 and below a*b/2^32 + p; callers track the bound.
R0H_HD uint32_t mul_lazy(uint32_t a, uint32_t b) {
  uint64_t t = (uint64_t)a * b;
  uint32_t m = (uint32_t)t * NPINV;
  return (uint32_t)((t + (uint64_t)m * P) >> 32);
}
R0H_HD uint32_t enc(uint32_t canonical) { return mul(canonical % P, R2); }
R0H_HD uint32_t dec(uint32_t a) { return mul(a, 1u); }
R0H_HD uint32_t fpow(uint32_t a, uint64_t n) {
  uint32_t r = ONE;
  while (n) {
    if (n & 1) r = mul(r, a);
    a = mul(a, a);
    n >>= 1;
  }
  return r;
}
R0H_HD uint32_t inv(uint32_t a) { return fpow(a, P - 2); }
// 11 in Montgomery form: 11 * 2^32 mod p
constexpr uint32_t BETA_M = (uint32_t)((11ull << 32) % P);

struct Fp4 {
  uint32_t e[4];
};
R0H_HD Fp4 fp4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { return Fp4{{a, b, c, d}}; }
R0H_HD Fp4 fp4_zero() { return Fp4{{0, 0, 0, 0}}; }
R0H_HD Fp4 fp4_one() { return Fp4{{ONE, 0, 0, 0}}; }
R0H_HD bool operator==(const Fp4& a, const Fp4& b) {
  return a.e[0] == b.e[0] && a.e[1] == b.e[1] && a.e[2] == b.e[2] && a.e[3] == b.e[3];
}
R0H_HD Fp4 operator+(const Fp4& a, const Fp4& b) {
  return Fp4{{add(a.e[0], b.e[0]), add(a.e[1], b.e[1]), add(a.e[2], b.e[2]), add(a.e[3], b.e[3])}};
}
R0H_HD Fp4 operator-(const Fp4& a, const Fp4& b) {
  return Fp4{{sub(a.e[0], b.e[0]), sub(a.e[1], b.e[1]), sub(a.e[2], b.e[2]), sub(a.e[3], b.e[3])}};
}
R0H_HD Fp4 scale(const Fp4& a, uint32_t s) { return Fp4{{mul(a.e[0], s), mul(a.e[1], s), mul(a.e[2], s), mul(a.e[3], s)}}; }
R0H_HD Fp4 operator*(const Fp4& a, const Fp4& b) {
  // (a0 + a1 x + a2 x^2 + a3 x^3)(b0 + ...), x^4 = 11; products are summed in 64 bits (at most four, 4 p^2 < 2^64)
  // and reduced once per coefficient
  typedef uint64_t u64;
  uint32_t c4 = reduce64((u64)a.e[1] * b.e[3] + (u64)a.e[2] * b.e[2] + (u64)a.e[3] * b.e[1]);
  uint32_t c5 = reduce64((u64)a.e[2] * b.e[3] + (u64)a.e[3] * b.e[2]);
  uint32_t c6 = reduce64((u64)a.e[3] * b.e[3]);
  Fp4 r;
  r.e[0] = reduce64((u64)a.e[0] * b.e[0] + (u64)BETA_M * c4);
  r.e[1] = reduce64((u64)a.e[0] * b.e[1] + (u64)a.e[1] * b.e[0] + (u64)BETA_M * c5);
  r.e[2] = reduce64((u64)a.e[0] * b.e[2] + (u64)a.e[1] * b.e[1] + (u64)a.e[2] * b.e[0] + (u64)BETA_M * c6);
  r.e[3] = reduce64((u64)a.e[0] * b.e[3] + (u64)a.e[1] * b.e[2] + (u64)a.e[2] * b.e[1] + (u64)a.e[3] * b.e[0]);
  return r;
}
R0H_HD Fp4 fp4_pow(Fp4 a, uint64_t n) {
  Fp4 r = fp4_one();
  while (n) {
    if (n & 1) r = r * a;
    a = a * a;
    n >>= 1;
  }
  return r;
}
// inverse through the tower Fp2[y]/(y^2 - x) over Fp[x]/(x^2 - 11): a = A + yB, a^-1 = (A - yB)/(A^2 - x B^2)
R0H_HD Fp4 fp4_inv(const Fp4& a) {
  uint32_t A0 = a.e[0], A1 = a.e[2], B0 = a.e[1], B1 = a.e[3];
  uint32_t A2_0 = add(mul(A0, A0), mul(BETA_M, mul(A1, A1))), A2_1 = mul(add(A0, A0), A1);
  uint32_t B2_0 = add(mul(B0, B0), mul(BETA_M, mul(B1, B1))), B2_1 = mul(add(B0, B0), B1);
  uint32_t D0 = sub(A2_0, mul(BETA_M, B2_1)), D1 = sub(A2_1, B2_0);
  uint32_t n = inv(sub(mul(D0, D0), mul(BETA_M, mul(D1, D1))));
  uint32_t I0 = mul(D0, n), I1 = neg(mul(D1, n));
  Fp4 r;
  r.e[0] = add(mul(A0, I0), mul(BETA_M, mul(A1, I1)));
  r.e[2] = add(mul(A0, I1), mul(A1, I0));
  r.e[1] = neg(add(mul(B0, I0), mul(BETA_M, mul(B1, I1))));
  r.e[3] = neg(add(mul(B0, I1), mul(B1, I0)));
  return r;
}

R0H_HD uint32_t bitrev(uint32_t x, uint32_t bits) {
#if defined(__HIP_DEVICE_COMPILE__)
  return bits ? __brev(x) >> (32 - bits) : 0u;
#else
  uint32_t r = 0;
  for (uint32_t i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
  return r;
#endif
}
inline uint32_t rou_fwd(uint32_t po2) { return fpow(enc(ROU_GEN), 1ull << (27 - po2)); }
inline uint32_t rou_rev(uint32_t po2) { return inv(rou_fwd(po2)); }

}  // namespace r0h
