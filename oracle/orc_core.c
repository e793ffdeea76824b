/* ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_field.h).  Core operations of the STARK pipeline on the CPU.
 *
 * Upstream modules restated (all un-vendored, recalled):
 *   a2-a5  risc0-zkp core/ntt.rs (`interpolate_ntt`, `evaluate_ntt`, `expand`, `bit_reverse`), hal/cpu.rs `zk_shift`
 *   a6-a7  risc0-zkp core/hash/poseidon2/mod.rs (`poseidon2_mix`, `unpadded_hash`, `hash_pair`), hal/cpu.rs
 *          `hash_rows`, `hash_fold`
 *   a8     risc0-zkp prove/merkle.rs / merkle.rs (`MerkleTreeParams`)
 *   a12-15 risc0-zkp hal/cpu.rs `batch_evaluate_any`, `mix_poly_coeffs`, `eltwise_sum_extelem`, `fri_fold`,
 *          `prefix_products`; core/poly.rs `poly_divide`, `poly_interpolate`
 *   a17    risc0-zkp core/hash/poseidon2/rng.rs
 */
#include "orc.h"
#include "orc_field.h"
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static int g_threads = 1;
void orc_set_threads(int n) {
  g_threads = n < 1 ? 1 : n;
#ifdef _OPENMP
  omp_set_num_threads(g_threads);
#endif
}
int orc_get_threads(void) { return g_threads; }

/* ------------------------------------------------------------------ field exports */
uint32_t orc_fp_mul(uint32_t a, uint32_t b) { return fp_mul(a, b); }
uint32_t orc_fp_enc(uint32_t c) { return fp_enc(c); }
uint32_t orc_fp_dec(uint32_t m) { return fp_dec(m); }
uint32_t orc_fp_inv(uint32_t a) { return fp_inv(a); }
uint32_t orc_fp_pow(uint32_t a, uint64_t n) { return fp_pow(a, n); }
void orc_fp4_mul(const uint32_t a[4], const uint32_t b[4], uint32_t out[4]) {
  fp4_t x, y; memcpy(&x, a, 16); memcpy(&y, b, 16);
  fp4_t r = fp4_mul(x, y); memcpy(out, &r, 16);
}
void orc_fp4_inv(const uint32_t a[4], uint32_t out[4]) {
  fp4_t x; memcpy(&x, a, 16);
  fp4_t r = fp4_inv(x); memcpy(out, &r, 16);
}

/* ROU_FWD[i] = 137^(2^(27-i)); ROU_REV[i] its inverse. */
uint32_t orc_rou_fwd(unsigned po2) { return fp_pow(fp_enc(ORC_ROU_GEN), 1ull << (ORC_MAX_ROU_PO2 - po2)); }
uint32_t orc_rou_rev(unsigned po2) { return fp_inv(orc_rou_fwd(po2)); }

/* ------------------------------------------------------------------ NTT */
static fp_t* twiddle_table(fp_t root, size_t half) {
  fp_t* t = (fp_t*)malloc(sizeof(fp_t) * (half ? half : 1));
  fp_t cur = ORC_ONE;
  for (size_t i = 0; i < half; i++) { t[i] = cur; cur = fp_mul(cur, root); }
  return t;
}

/* natural-order evaluations on {w^i} -> bit-reversed coefficients (decimation in frequency, inverse roots, 1/N) */
static void interpolate_one(fp_t* io, unsigned n, const fp_t* tw /* ROU_REV[n]^i */, fp_t norm) {
  size_t size = (size_t)1 << n;
  for (unsigned s = n; s >= 1; s--) {
    size_t half = (size_t)1 << (s - 1), stride = (size_t)1 << (n - s);
    for (size_t blk = 0; blk < size; blk += 2 * half)
      for (size_t i = 0; i < half; i++) {
        fp_t a = io[blk + i], b = io[blk + i + half];
        io[blk + i] = fp_add(a, b);
        io[blk + i + half] = fp_mul(fp_sub(a, b), tw[i * stride]);
      }
  }
  for (size_t i = 0; i < size; i++) io[i] = fp_mul(io[i], norm);
}

void orc_batch_interpolate_ntt(uint32_t* io, uint32_t count, uint32_t po2) {
  size_t size = (size_t)1 << po2;
  fp_t* tw = twiddle_table(orc_rou_rev(po2), size / 2);
  fp_t norm = fp_inv(fp_enc((uint32_t)size));
#pragma omp parallel for schedule(dynamic)
  for (uint32_t c = 0; c < count; c++) interpolate_one(io + (size_t)c * size, po2, tw, norm);
  free(tw);
}

/* bit-reversed coefficients (size 2^in_po2) -> natural-order evaluations on the 2^(in_po2+expand_bits) domain.
 * Zero-padding the high coefficients equals repeating each bit-reversed entry 2^expand_bits times and skipping the
 * first expand_bits butterfly layers. */
static void evaluate_one(fp_t* out, const fp_t* in, unsigned n, unsigned expand_bits, const fp_t* tw /* ROU_FWD[n]^i */) {
  size_t size = (size_t)1 << n;
  for (size_t i = 0; i < size; i++) out[i] = in[i >> expand_bits];
  for (unsigned s = expand_bits + 1; s <= n; s++) {
    size_t half = (size_t)1 << (s - 1), stride = (size_t)1 << (n - s);
    for (size_t blk = 0; blk < size; blk += 2 * half)
      for (size_t i = 0; i < half; i++) {
        fp_t a = out[blk + i], b = fp_mul(out[blk + i + half], tw[i * stride]);
        out[blk + i] = fp_add(a, b);
        out[blk + i + half] = fp_sub(a, b);
      }
  }
}

void orc_batch_expand_into_evaluate_ntt(uint32_t* out, const uint32_t* in, uint32_t count, uint32_t in_po2,
                                        uint32_t expand_bits) {
  unsigned n = in_po2 + expand_bits;
  size_t in_size = (size_t)1 << in_po2, out_size = (size_t)1 << n;
  fp_t* tw = twiddle_table(orc_rou_fwd(n), out_size / 2);
#pragma omp parallel for schedule(dynamic)
  for (uint32_t c = 0; c < count; c++) evaluate_one(out + (size_t)c * out_size, in + (size_t)c * in_size, n, expand_bits, tw);
  free(tw);
}

void orc_batch_bit_reverse(uint32_t* io, uint32_t count, uint32_t po2) {
  size_t size = (size_t)1 << po2;
#pragma omp parallel for
  for (uint32_t c = 0; c < count; c++) {
    fp_t* col = io + (size_t)c * size;
    for (size_t i = 0; i < size; i++) {
      size_t j = orc_bitrev((uint32_t)i, po2);
      if (i < j) { fp_t t = col[i]; col[i] = col[j]; col[j] = t; }
    }
  }
}

/* f(x) -> f(3x) on bit-reversed coefficients: entry at position i holds the coefficient of x^brev(i). */
void orc_zk_shift(uint32_t* io, uint32_t count, uint32_t po2) {
  size_t size = (size_t)1 << po2;
  fp_t* pow3 = twiddle_table(fp_enc(3), size);
#pragma omp parallel for
  for (uint32_t c = 0; c < count; c++) {
    fp_t* col = io + (size_t)c * size;
    for (size_t i = 0; i < size; i++) col[i] = fp_mul(col[i], pow3[orc_bitrev((uint32_t)i, po2)]);
  }
  free(pow3);
}

/* ------------------------------------------------------------------ Poseidon2 */
#define ROUNDS_HALF_FULL 4
#define ROUNDS_PARTIAL 21
#define N_ROUNDS (2 * ROUNDS_HALF_FULL + ROUNDS_PARTIAL)
static fp_t g_rc[ORC_CELLS * N_ROUNDS]; /* Montgomery */
static fp_t g_diag[ORC_CELLS];          /* Montgomery, (mu_i - 1) */
static uint32_t g_rc_canon[ORC_CELLS * N_ROUNDS], g_diag_canon[ORC_CELLS];
static int g_p2_ready = 0;

/* Grain LFSR, self-shrinking mode, exactly as the Poseidon papers' parameter script (independent of the python tool). */
typedef struct { uint8_t s[80]; unsigned head; } grain_t;
static int grain_step(grain_t* g) {
  unsigned h = g->head;
#define GB(k) g->s[(h + (k)) % 80]
  uint8_t b = GB(62) ^ GB(51) ^ GB(38) ^ GB(23) ^ GB(13) ^ GB(0);
#undef GB
  g->s[h] = b;
  g->head = (h + 1) % 80;
  return b;
}
static int grain_bit(grain_t* g) {
  int b = grain_step(g);
  while (b == 0) { grain_step(g); b = grain_step(g); }
  return grain_step(g);
}
static uint32_t grain_raw(grain_t* g, unsigned nbits) {
  uint32_t v = 0;
  for (unsigned i = 0; i < nbits; i++) v = (v << 1) | (uint32_t)grain_bit(g);
  return v;
}
static void put_bits(uint8_t* s, unsigned* pos, uint32_t v, unsigned len) {
  for (unsigned i = 0; i < len; i++) s[(*pos)++] = (v >> (len - 1 - i)) & 1u;
}
static void poseidon2_setup(void) {
  if (g_p2_ready) return;
  grain_t g;
  unsigned pos = 0;
  put_bits(g.s, &pos, 1, 2); put_bits(g.s, &pos, 0, 4); put_bits(g.s, &pos, 31, 12); put_bits(g.s, &pos, ORC_CELLS, 12);
  put_bits(g.s, &pos, 2 * ROUNDS_HALF_FULL, 10); put_bits(g.s, &pos, ROUNDS_PARTIAL, 10);
  while (pos < 80) g.s[pos++] = 1;
  g.head = 0;
  for (int i = 0; i < 160; i++) grain_step(&g);
  memset(g_rc_canon, 0, sizeof g_rc_canon);
  for (int r = 0; r < N_ROUNDS; r++) {
    int full = r < ROUNDS_HALF_FULL || r >= ROUNDS_HALF_FULL + ROUNDS_PARTIAL;
    int lanes = full ? ORC_CELLS : 1;
    for (int i = 0; i < lanes; i++) {
      uint32_t v;
      do { v = grain_raw(&g, 31); } while (v >= ORC_P);
      g_rc_canon[r * ORC_CELLS + i] = v;
    }
  }
  for (int i = 0; i < 4 * ORC_CELLS; i++) grain_raw(&g, 31); /* rejected candidate diagonals */
  for (int i = 0; i < ORC_CELLS; i++) g_diag_canon[i] = (grain_raw(&g, 31) % ORC_P + ORC_P - 1) % ORC_P;
  for (int i = 0; i < ORC_CELLS * N_ROUNDS; i++) g_rc[i] = fp_enc(g_rc_canon[i]);
  for (int i = 0; i < ORC_CELLS; i++) g_diag[i] = fp_enc(g_diag_canon[i]);
  g_p2_ready = 1;
}
void orc_poseidon2_consts(uint32_t* rc, uint32_t* diag) {
  poseidon2_setup();
  memcpy(rc, g_rc_canon, sizeof g_rc_canon);
  memcpy(diag, g_diag_canon, sizeof g_diag_canon);
}

static inline fp_t sbox7(fp_t x) {
  fp_t x2 = fp_mul(x, x), x4 = fp_mul(x2, x2);
  return fp_mul(fp_mul(x4, x2), x);
}
/* external layer: circ(2*M4, M4, ..., M4) with M4 = [[5,7,1,3],[4,6,1,1],[1,3,5,7],[1,1,4,6]] (Poseidon2 paper, app. B) */
static void m_ext(fp_t* c) {
  fp_t col[4] = {0, 0, 0, 0};
  for (int k = 0; k < ORC_CELLS; k += 4) {
    fp_t a = c[k], b = c[k + 1], d = c[k + 2], e = c[k + 3];
    fp_t t0 = fp_add(a, b), t1 = fp_add(d, e);
    fp_t t2 = fp_add(fp_add(b, b), t1), t3 = fp_add(fp_add(e, e), t0);
    fp_t t1x4 = fp_add(fp_add(t1, t1), fp_add(t1, t1)), t0x4 = fp_add(fp_add(t0, t0), fp_add(t0, t0));
    fp_t t4 = fp_add(t1x4, t3), t5 = fp_add(t0x4, t2);
    c[k] = fp_add(t3, t5); c[k + 1] = t5; c[k + 2] = fp_add(t2, t4); c[k + 3] = t4;
    for (int j = 0; j < 4; j++) col[j] = fp_add(col[j], c[k + j]);
  }
  for (int i = 0; i < ORC_CELLS; i++) c[i] = fp_add(c[i], col[i & 3]);
}
static void m_int(fp_t* c) {
  fp_t sum = 0;
  for (int i = 0; i < ORC_CELLS; i++) sum = fp_add(sum, c[i]);
  for (int i = 0; i < ORC_CELLS; i++) c[i] = fp_add(sum, fp_mul(g_diag[i], c[i]));
}
static void p2_mix(fp_t* c) {
  int r = 0;
  m_ext(c);
  for (int k = 0; k < ROUNDS_HALF_FULL; k++, r++) {
    for (int i = 0; i < ORC_CELLS; i++) c[i] = sbox7(fp_add(c[i], g_rc[r * ORC_CELLS + i]));
    m_ext(c);
  }
  for (int k = 0; k < ROUNDS_PARTIAL; k++, r++) {
    c[0] = sbox7(fp_add(c[0], g_rc[r * ORC_CELLS]));
    m_int(c);
  }
  for (int k = 0; k < ROUNDS_HALF_FULL; k++, r++) {
    for (int i = 0; i < ORC_CELLS; i++) c[i] = sbox7(fp_add(c[i], g_rc[r * ORC_CELLS + i]));
    m_ext(c);
  }
}
void orc_poseidon2_mix(uint32_t cells[ORC_CELLS]) { poseidon2_setup(); p2_mix(cells); }

/* Sponge in overwrite mode, rate 16, zero padding of the last partial block; empty input costs one permutation. */
static void sponge_strided(const fp_t* in, size_t n, size_t stride, fp_t digest[8]) {
  fp_t st[ORC_CELLS];
  memset(st, 0, sizeof st);
  size_t used = 0;
  for (size_t i = 0; i < n; i++) {
    st[used++] = in[i * stride];
    if (used == ORC_RATE) { p2_mix(st); used = 0; }
  }
  if (used != 0 || n == 0) {
    for (size_t i = used; i < ORC_RATE; i++) st[i] = 0;
    p2_mix(st);
  }
  memcpy(digest, st, 8 * sizeof(fp_t));
}
void orc_hash_elem_slice(const uint32_t* elems, size_t n, uint32_t digest[8]) {
  poseidon2_setup();
  sponge_strided(elems, n, 1, digest);
}
/* The sponge above laid out as the rows of the recursion circuit's in-circuit hash (tools/sponge_component.py; the constraints are in
 * the circuit blob): 30 rows per permutation -- absorb + external layer, then one round per row -- and 65 columns of n = 2^po2 rows:
 * st[24] the state after the row's step, aux[24] the cubes (x + rc)^3 of the lanes the round's S-box touches, in[16] the absorbed
 * words on a permutation's first row, act = 1 on the sponge's rows.  Returns 0, or -1 when the words do not fit the trace. */
int orc_sponge_trace(const uint32_t* words, size_t n_words, uint32_t po2, uint32_t* cols) {
  poseidon2_setup();
  const size_t n = (size_t)1 << po2, n_perm = n_words ? (n_words + ORC_RATE - 1) / ORC_RATE : 1;
  if (n_perm * 30 >= n) return -1;
  memset(cols, 0, 65 * n * sizeof(uint32_t));
  fp_t st[ORC_CELLS];
  memset(st, 0, sizeof st);
  size_t row = 0;
#define COL(k) (cols + (size_t)(k) * n)
  for (size_t q = 0; q < n_perm; q++) {
    for (size_t j = 0; j < ORC_RATE; j++) {
      st[j] = q * ORC_RATE + j < n_words ? words[q * ORC_RATE + j] : 0;
      COL(48 + j)[row] = st[j];
    }
    m_ext(st);
    for (int j = 0; j < ORC_CELLS; j++) COL(j)[row] = st[j];
    COL(64)[row++] = ORC_ONE;
    for (int r = 0; r < 2 * ROUNDS_HALF_FULL + ROUNDS_PARTIAL; r++, row++) {
      const int full = r < ROUNDS_HALF_FULL || r >= ROUNDS_HALF_FULL + ROUNDS_PARTIAL;
      for (int j = 0; j < (full ? ORC_CELLS : 1); j++) {
        fp_t t = fp_add(st[j], g_rc[r * ORC_CELLS + j]), cube = fp_mul(fp_mul(t, t), t);
        COL(24 + j)[row] = cube;
        st[j] = fp_mul(fp_mul(cube, cube), t);
      }
      if (full) m_ext(st); else m_int(st);
      for (int j = 0; j < ORC_CELLS; j++) COL(j)[row] = st[j];
      COL(64)[row] = ORC_ONE;
    }
  }
#undef COL
  return 0;
}
void orc_hash_pair(const uint32_t a[8], const uint32_t b[8], uint32_t out[8]) {
  poseidon2_setup();
  fp_t st[ORC_CELLS];
  memcpy(st, a, 32); memcpy(st + 8, b, 32); memset(st + 16, 0, 32);
  p2_mix(st);
  memcpy(out, st, 32);
}
void orc_hash_rows(uint32_t* digests, const uint32_t* matrix, size_t rows, size_t cols) {
  poseidon2_setup();
#pragma omp parallel for schedule(static)
  for (size_t r = 0; r < rows; r++) sponge_strided(matrix + r, cols, rows, digests + r * 8);
}
void orc_hash_fold(uint32_t* nodes, size_t output_size) {
  poseidon2_setup();
#pragma omp parallel for schedule(static)
  for (size_t i = output_size; i < 2 * output_size; i++) orc_hash_pair(nodes + 2 * i * 8, nodes + (2 * i + 1) * 8, nodes + i * 8);
}

/* ------------------------------------------------------------------ Merkle */
void orc_merkle_params(orc_merkle_params_t* p, size_t row_size, size_t col_size, size_t queries) {
  p->row_size = row_size; p->col_size = col_size; p->queries = queries;
  size_t layers = 0;
  while (((size_t)1 << layers) < row_size) layers++;
  p->layers = layers;
  p->top_layer = 0;
  for (size_t i = 1; i < layers; i++) {
    if (((size_t)1 << i) > queries) break;
    p->top_layer = i;
  }
  p->top_size = (size_t)1 << p->top_layer;
}
void orc_merkle_build(uint32_t* nodes, const uint32_t* matrix, size_t row_size, size_t col_size) {
  orc_hash_rows(nodes + row_size * 8, matrix, row_size, col_size);
  for (size_t sz = row_size / 2; sz >= 1; sz /= 2) orc_hash_fold(nodes, sz);
}

/* ------------------------------------------------------------------ streaming ops */
static inline fp4_t ld4(const uint32_t* p) { fp4_t r; memcpy(&r, p, 16); return r; }
static inline void st4(uint32_t* p, fp4_t v) { memcpy(p, &v, 16); }

/* out[k] = sum_i coeffs[which[k]][i] * xs[k]^i  (natural-order base-field coefficients, extension point) */
void orc_batch_evaluate_any(const uint32_t* coeffs, uint32_t po2, const uint32_t* which, const uint32_t* xs,
                            uint32_t n_eval, uint32_t* out) {
  size_t n = (size_t)1 << po2;
#pragma omp parallel for schedule(dynamic)
  for (uint32_t k = 0; k < n_eval; k++) {
    const fp_t* poly = coeffs + (size_t)which[k] * n;
    fp4_t x = ld4(xs + 4 * k), tot = fp4_zero();
    for (size_t i = n; i-- > 0;) tot = fp4_add(fp4_mul(tot, x), fp4_from_fp(poly[i]));
    st4(out + 4 * k, tot);
  }
}

/* combos[combo_of[c]][i] += mix_start * mix^c * input[c][i] */
void orc_mix_poly_coeffs(uint32_t* combos, const uint32_t mix_start[4], const uint32_t mix[4], const uint32_t* input,
                         const uint32_t* combo_of, uint32_t input_count, uint32_t po2) {
  size_t n = (size_t)1 << po2;
  fp4_t* pw = (fp4_t*)malloc(sizeof(fp4_t) * (input_count ? input_count : 1));
  fp4_t cur = ld4(mix_start), m = ld4(mix);
  for (uint32_t c = 0; c < input_count; c++) { pw[c] = cur; cur = fp4_mul(cur, m); }
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n; i++)
    for (uint32_t c = 0; c < input_count; c++) {
      uint32_t* dst = combos + ((size_t)combo_of[c] * n + i) * 4;
      st4(dst, fp4_add(ld4(dst), fp4_scale(pw[c], input[(size_t)c * n + i])));
    }
  free(pw);
}

void orc_eltwise_sum_extelem(uint32_t* out, const uint32_t* in, uint32_t count, uint32_t n) {
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n; i++) {
    fp4_t tot = fp4_zero();
    for (uint32_t c = 0; c < count; c++) tot = fp4_add(tot, ld4(in + ((size_t)c * n + i) * 4));
    for (int k = 0; k < 4; k++) out[(size_t)k * n + i] = tot.e[k];
  }
}

/* Fold-by-16 on bit-reversed coefficients: out[idx] = sum_j mix^j * in[brev4(j)*n_out + idx] */
void orc_fri_fold(uint32_t* out, const uint32_t* in, const uint32_t mix[4], uint32_t n_out) {
  size_t n_in = (size_t)n_out * ORC_FRI_FOLD;
  fp4_t m = ld4(mix);
#pragma omp parallel for schedule(static)
  for (size_t idx = 0; idx < n_out; idx++) {
    fp4_t tot = fp4_zero(), cur = fp4_one();
    for (uint32_t j = 0; j < ORC_FRI_FOLD; j++) {
      size_t src = (size_t)orc_bitrev(j, 4) * n_out + idx;
      fp4_t v = {{in[src], in[n_in + src], in[2 * n_in + src], in[3 * n_in + src]}};
      tot = fp4_add(tot, fp4_mul(cur, v));
      cur = fp4_mul(cur, m);
    }
    for (int k = 0; k < 4; k++) out[(size_t)k * n_out + idx] = tot.e[k];
  }
}

void orc_prefix_products(uint32_t* io, uint32_t n) {
  fp4_t cur = fp4_one();
  for (size_t i = 0; i < n; i++) { cur = fp4_mul(cur, ld4(io + 4 * i)); st4(io + 4 * i, cur); }
}

/* synthetic division by (x - z): poly becomes the quotient (top coefficient 0), remainder returned */
void orc_poly_divide(uint32_t* poly, uint32_t n, const uint32_t z[4], uint32_t rem[4]) {
  fp4_t zz = ld4(z), cur = fp4_zero();
  for (size_t i = n; i-- > 0;) {
    fp4_t next = fp4_add(fp4_mul(zz, cur), ld4(poly + 4 * i));
    st4(poly + 4 * i, cur);
    cur = next;
  }
  st4(rem, cur);
}

/* Lagrange interpolation through (xs[i], ys[i]), i < n (n is at most a handful): coefficients low to high. */
void orc_poly_interpolate(uint32_t* out, const uint32_t* xs, const uint32_t* ys, uint32_t n) {
  fp4_t acc[16], basis[17];
  for (uint32_t i = 0; i < n; i++) acc[i] = fp4_zero();
  for (uint32_t i = 0; i < n; i++) {
    /* basis = prod_{j != i} (x - xs[j]) */
    uint32_t deg = 0;
    basis[0] = fp4_one();
    fp4_t denom = fp4_one(), xi = ld4(xs + 4 * i);
    for (uint32_t j = 0; j < n; j++) {
      if (j == i) continue;
      fp4_t xj = ld4(xs + 4 * j);
      basis[deg + 1] = fp4_zero();
      for (uint32_t k = deg + 1; k-- > 0;) {
        basis[k + 1] = fp4_add(basis[k + 1], basis[k]);
        basis[k] = fp4_sub(fp4_zero(), fp4_mul(basis[k], xj));
      }
      deg++;
      denom = fp4_mul(denom, fp4_sub(xi, xj));
    }
    fp4_t scale = fp4_mul(ld4(ys + 4 * i), fp4_inv(denom));
    for (uint32_t k = 0; k < n; k++) acc[k] = fp4_add(acc[k], fp4_mul(basis[k], scale));
  }
  for (uint32_t k = 0; k < n; k++) st4(out + 4 * k, acc[k]);
}

/* ------------------------------------------------------------------ transcript RNG */
void orc_rng_init(orc_rng_t* r) { poseidon2_setup(); memset(r, 0, sizeof *r); }
void orc_rng_mix(orc_rng_t* r, const uint32_t digest[8]) {
  if (r->pool_used != 0) { p2_mix(r->cells); r->pool_used = 0; }
  for (int i = 0; i < 8; i++) r->cells[i] = fp_add(r->cells[i], digest[i] % ORC_P);
  p2_mix(r->cells);
}
uint32_t orc_rng_elem(orc_rng_t* r) {
  if (r->pool_used == ORC_RATE) { p2_mix(r->cells); r->pool_used = 0; }
  return r->cells[r->pool_used++];
}
uint32_t orc_rng_bits(orc_rng_t* r, uint32_t bits) {
  uint32_t val = fp_dec(orc_rng_elem(r));
  for (int i = 0; i < 3; i++) {
    uint32_t nv = fp_dec(orc_rng_elem(r));
    if (val == 0) val = nv;
  }
  return val & (uint32_t)(((uint64_t)1 << bits) - 1);
}
