/* ORACLE -- TEST INFRASTRUCTURE ONLY.  Not shipped, not linked into the product library.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
 *
 * BabyBear field arithmetic, CPU restatement.
 *
 * Follows (un-vendored third party): risc0-zkp 3.0.4 `field/baby_bear.rs`
 * (Elem = u32 in Montgomery form, R = 2^32; ExtElem = Fp[x]/(x^4 - 11)) and risc0-sys 1.5.0 `fp.h`/`fpext.h`.
 * Reference call site that reaches it: host/src/main.rs:423 (prover.prove) and verifier/src/main.rs:124-126.
 *
 * Pinned by: the 28-entry roots-of-unity table recalled from risc0 equals 137^(2^(27-i)) mod p for every i
 * (tests/test_oracle_field.py), plus exhaustive-ish algebraic identities.  No golden vector for this path
 * exists in /root/reference.
 */
#ifndef ORC_FIELD_H
#define ORC_FIELD_H
#include <stdint.h>

#define ORC_P 2013265921u            /* 15 * 2^27 + 1 */
#define ORC_NPINV 0x77ffffffu        /* -p^-1 mod 2^32 */
#define ORC_R2 1172168163u           /* 2^64 mod p */
#define ORC_ONE 268435454u           /* 2^32 mod p: Montgomery form of 1 */
#define ORC_ROU_GEN 137u             /* primitive 2^27-th root of unity (canonical) */
#define ORC_MAX_ROU_PO2 27
#define ORC_BETA 11u                 /* x^4 = BETA in the extension (canonical) */

typedef uint32_t fp_t;               /* Montgomery word, always < p */
typedef struct { fp_t e[4]; } fp4_t;

static inline fp_t fp_add(fp_t a, fp_t b) { uint32_t s = a + b; return s >= ORC_P ? s - ORC_P : s; }
static inline fp_t fp_sub(fp_t a, fp_t b) { return a >= b ? a - b : a + ORC_P - b; }
static inline fp_t fp_neg(fp_t a) { return a ? ORC_P - a : 0; }
/* Montgomery product: a*b*2^-32 mod p */
static inline fp_t fp_mul(fp_t a, fp_t b) {
  uint64_t t = (uint64_t)a * b;
  uint32_t m = (uint32_t)t * ORC_NPINV;
  uint32_t r = (uint32_t)((t + (uint64_t)m * ORC_P) >> 32);
  return r >= ORC_P ? r - ORC_P : r;
}
static inline fp_t fp_enc(uint32_t canonical) { return fp_mul(canonical % ORC_P, ORC_R2); }
static inline uint32_t fp_dec(fp_t a) { return fp_mul(a, 1u); }
static inline fp_t fp_pow(fp_t a, uint64_t n) {
  fp_t r = ORC_ONE;
  while (n) { if (n & 1) r = fp_mul(r, a); a = fp_mul(a, a); n >>= 1; }
  return r;
}
static inline fp_t fp_inv(fp_t a) { return fp_pow(a, ORC_P - 2); }

static inline fp4_t fp4_zero(void) { fp4_t r = {{0, 0, 0, 0}}; return r; }
static inline fp4_t fp4_one(void) { fp4_t r = {{ORC_ONE, 0, 0, 0}}; return r; }
static inline fp4_t fp4_from_fp(fp_t a) { fp4_t r = {{a, 0, 0, 0}}; return r; }
static inline int fp4_eq(fp4_t a, fp4_t b) {
  return a.e[0] == b.e[0] && a.e[1] == b.e[1] && a.e[2] == b.e[2] && a.e[3] == b.e[3];
}
static inline fp4_t fp4_add(fp4_t a, fp4_t b) {
  fp4_t r; for (int i = 0; i < 4; i++) r.e[i] = fp_add(a.e[i], b.e[i]); return r;
}
static inline fp4_t fp4_sub(fp4_t a, fp4_t b) {
  fp4_t r; for (int i = 0; i < 4; i++) r.e[i] = fp_sub(a.e[i], b.e[i]); return r;
}
static inline fp4_t fp4_scale(fp4_t a, fp_t s) {
  fp4_t r; for (int i = 0; i < 4; i++) r.e[i] = fp_mul(a.e[i], s); return r;
}
/* schoolbook product then fold x^4 -> 11, x^5 -> 11x, x^6 -> 11x^2 */
static inline fp4_t fp4_mul(fp4_t a, fp4_t b) {
  fp_t c[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) c[i + j] = fp_add(c[i + j], fp_mul(a.e[i], b.e[j]));
  const fp_t beta = fp_enc(ORC_BETA);
  fp4_t r;
  r.e[0] = fp_add(c[0], fp_mul(beta, c[4]));
  r.e[1] = fp_add(c[1], fp_mul(beta, c[5]));
  r.e[2] = fp_add(c[2], fp_mul(beta, c[6]));
  r.e[3] = c[3];
  return r;
}
static inline fp4_t fp4_pow(fp4_t a, uint64_t n) {
  fp4_t r = fp4_one();
  while (n) { if (n & 1) r = fp4_mul(r, a); a = fp4_mul(a, a); n >>= 1; }
  return r;
}
/* Inverse through the tower Fp4 = Fp2[y]/(y^2 - x), Fp2 = Fp[x]/(x^2 - 11):
 * a = A + yB with A = a0 + a2 x, B = a1 + a3 x;  a^-1 = (A - yB) / (A^2 - x B^2). */
static inline fp4_t fp4_inv(fp4_t a) {
  const fp_t beta = fp_enc(ORC_BETA);
  fp_t A0 = a.e[0], A1 = a.e[2], B0 = a.e[1], B1 = a.e[3];
  /* A^2 = (A0^2 + 11 A1^2) + (2 A0 A1) x ; B^2 likewise ; x*B^2 = 11*B2_1 + B2_0 x */
  fp_t A2_0 = fp_add(fp_mul(A0, A0), fp_mul(beta, fp_mul(A1, A1)));
  fp_t A2_1 = fp_mul(fp_add(A0, A0), A1);
  fp_t B2_0 = fp_add(fp_mul(B0, B0), fp_mul(beta, fp_mul(B1, B1)));
  fp_t B2_1 = fp_mul(fp_add(B0, B0), B1);
  fp_t D0 = fp_sub(A2_0, fp_mul(beta, B2_1));
  fp_t D1 = fp_sub(A2_1, B2_0);
  /* (D0 + D1 x)^-1 = (D0 - D1 x) / (D0^2 - 11 D1^2) */
  fp_t n = fp_inv(fp_sub(fp_mul(D0, D0), fp_mul(beta, fp_mul(D1, D1))));
  fp_t I0 = fp_mul(D0, n), I1 = fp_neg(fp_mul(D1, n));
  /* (A0 + A1 x)(I0 + I1 x) and -(B0 + B1 x)(I0 + I1 x) */
  fp4_t r;
  r.e[0] = fp_add(fp_mul(A0, I0), fp_mul(beta, fp_mul(A1, I1)));
  r.e[2] = fp_add(fp_mul(A0, I1), fp_mul(A1, I0));
  r.e[1] = fp_neg(fp_add(fp_mul(B0, I0), fp_mul(beta, fp_mul(B1, I1))));
  r.e[3] = fp_neg(fp_add(fp_mul(B0, I1), fp_mul(B1, I0)));
  return r;
}

static inline uint32_t orc_bitrev(uint32_t x, unsigned bits) {
  uint32_t r = 0;
  for (unsigned i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
  return r;
}
#endif
