/* ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_field.h).  The segment-proof sequencer and its verifier on the CPU.
 *
 * Upstream modules restated (un-vendored, recalled):
 *   risc0-zkp prove/prover.rs   `Prover::{new, commit_group, finalize}`, `make_coeffs`
 *   risc0-zkp prove/poly_group.rs `PolyGroup::new` (expand+evaluate, then bit-reverse coeffs to natural order)
 *   risc0-zkp prove/merkle.rs   `MerkleTreeProver::{new, commit, prove}`
 *   risc0-zkp prove/fri.rs      `fri_prove`, `ProveRoundInfo`
 *   risc0-zkp prove/write_iop.rs, verify/read_iop.rs
 *   risc0-zkp verify/mod.rs     `Verifier::verify`, `fri_eval_taps`, `compute_combos`; verify/fri.rs `fri_verify`
 *   risc0-circuit-rv32im 4.0.4 prove/hal/mod.rs (segment driver: seed transcript, commit CODE+DATA, draw mix, accum, finalize)
 * The verifier is what verifier/src/main.rs:124-126 (`receipt.verify`) runs for every segment seal.
 *
 * Seal word order (PARITY UNPINNED against risc0 -- no real seal exists under /root/reference):
 *   globals ++ [po2] | CODE top | DATA top | ACCUM top | CHECK top | coeff_u | FRI round tops... | final coeffs |
 *   50 x { ACCUM, CODE, DATA, CHECK column+path ; per FRI round column+path }
 */
#include "orc_circuit.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline fp4_t ld4(const uint32_t* p) { fp4_t r; memcpy(&r, p, 16); return r; }
static inline void st4(uint32_t* p, fp4_t v) { memcpy(p, &v, 16); }
static unsigned log2u(size_t x) { unsigned n = 0; while (((size_t)1 << n) < x) n++; return n; }

/* ------------------------------------------------------------------ write side of the IOP */
typedef struct { uint32_t* w; size_t n, cap; orc_rng_t rng; } wiop_t;
static void wiop_write(wiop_t* io, const uint32_t* src, size_t n) {
  if (io->n + n > io->cap) {
    while (io->n + n > io->cap) io->cap = io->cap ? io->cap * 2 : 4096;
    io->w = (uint32_t*)realloc(io->w, io->cap * 4);
  }
  memcpy(io->w + io->n, src, n * 4);
  io->n += n;
}
static fp4_t rng_ext(orc_rng_t* r) { fp4_t v; for (int i = 0; i < 4; i++) v.e[i] = orc_rng_elem(r); return v; }

/* risc0-circuit-rv32im prove/hal: the transcript starts with the hashes of two 16-byte ProtocolInfo tags, each byte one
 * field element: the proof system's ("RISC0_STARK:v1__") and the circuit's (from the blob). */
static void transcript_seed(orc_rng_t* rng, const orc_circuit_t* c) {
  static const char proof_system_info[] = "RISC0_STARK:v1__";
  uint32_t e[16], d[8];
  for (int i = 0; i < 16; i++) e[i] = fp_enc((uint8_t)proof_system_info[i]);
  orc_hash_elem_slice(e, 16, d);
  orc_rng_mix(rng, d);
  for (int i = 0; i < 16; i++) e[i] = fp_enc(c->info[i]);
  orc_hash_elem_slice(e, 16, d);
  orc_rng_mix(rng, d);
}

/* ------------------------------------------------------------------ Merkle prover */
typedef struct { orc_merkle_params_t p; uint32_t* nodes; const uint32_t* matrix; } merkle_t;
static void merkle_new(merkle_t* m, const uint32_t* matrix, size_t rows, size_t cols) {
  orc_merkle_params(&m->p, rows, cols, ORC_QUERIES);
  m->nodes = (uint32_t*)malloc(rows * 2 * 32);
  memset(m->nodes, 0, 64);
  m->matrix = matrix;
  orc_merkle_build(m->nodes, matrix, rows, cols);
}
static void merkle_commit(merkle_t* m, wiop_t* io) {
  wiop_write(io, m->nodes + m->p.top_size * 8, m->p.top_size * 8);
  orc_rng_mix(&io->rng, m->nodes + 8);
}
static void merkle_prove(merkle_t* m, wiop_t* io, size_t idx) {
  for (size_t i = 0; i < m->p.col_size; i++) wiop_write(io, &m->matrix[i * m->p.row_size + idx], 1);
  idx += m->p.row_size;
  while (idx >= 2 * m->p.top_size) {
    wiop_write(io, m->nodes + (idx ^ 1) * 8, 8);
    idx /= 2;
  }
}

/* ------------------------------------------------------------------ poly groups */
typedef struct { uint32_t* coeffs; uint32_t* evaluated; uint32_t count; merkle_t merkle; } group_t;
/* coeffs: bit-reversed, already in the shifted variable.  Evaluate on 4N, commit, then put coeffs in natural order. */
static void group_finish(group_t* g, uint32_t po2) {
  size_t n = (size_t)1 << po2;
  g->evaluated = (uint32_t*)malloc((size_t)g->count * n * ORC_INV_RATE * 4);
  orc_batch_expand_into_evaluate_ntt(g->evaluated, g->coeffs, g->count, po2, 2);
  orc_batch_bit_reverse(g->coeffs, g->count, po2);
  merkle_new(&g->merkle, g->evaluated, n * ORC_INV_RATE, g->count);
}
static void group_from_witness(group_t* g, const uint32_t* wit, uint32_t count, uint32_t po2) {
  size_t n = (size_t)1 << po2;
  g->count = count;
  g->coeffs = (uint32_t*)malloc((size_t)count * n * 4);
  memcpy(g->coeffs, wit, (size_t)count * n * 4);
  orc_batch_interpolate_ntt(g->coeffs, count, po2);
  orc_zk_shift(g->coeffs, count, po2);
  group_finish(g, po2);
}
static void group_free(group_t* g) { free(g->coeffs); free(g->evaluated); free(g->merkle.nodes); }

void orc_code_root(const uint32_t* code, uint32_t count, uint32_t po2, uint32_t root[8]) {
  group_t g;
  group_from_witness(&g, code, count, po2);
  memcpy(root, g.merkle.nodes + 8, 32);
  group_free(&g);
}

/* ------------------------------------------------------------------ FRI prover */
typedef struct { size_t domain; uint32_t* evaluated; merkle_t merkle; } fri_round_t;

static void fri_prove(wiop_t* io, uint32_t* coeffs /* [4][n] bit-reversed, consumed */, size_t n, group_t** groups, int n_groups) {
  size_t orig_domain = n * ORC_INV_RATE;
  fri_round_t rounds[8];
  int n_rounds = 0;
  while (n > ORC_FRI_MIN_DEGREE) {
    fri_round_t* r = &rounds[n_rounds++];
    r->domain = n * ORC_INV_RATE;
    r->evaluated = (uint32_t*)malloc(r->domain * 16);
    orc_batch_expand_into_evaluate_ntt(r->evaluated, coeffs, 4, log2u(n), 2);
    merkle_new(&r->merkle, r->evaluated, r->domain / ORC_FRI_FOLD, ORC_FRI_FOLD * 4);
    merkle_commit(&r->merkle, io);
    fp4_t fold_mix = rng_ext(&io->rng);
    uint32_t* out = (uint32_t*)malloc(n / ORC_FRI_FOLD * 16);
    orc_fri_fold(out, coeffs, fold_mix.e, (uint32_t)(n / ORC_FRI_FOLD));
    free(coeffs);
    coeffs = out;
    n /= ORC_FRI_FOLD;
  }
  orc_batch_bit_reverse(coeffs, 4, log2u(n));
  wiop_write(io, coeffs, 4 * n);
  uint32_t d[8];
  orc_hash_elem_slice(coeffs, 4 * n, d);
  orc_rng_mix(&io->rng, d);
  free(coeffs);
  for (int q = 0; q < ORC_QUERIES; q++) {
    size_t pos = orc_rng_bits(&io->rng, log2u(orig_domain)) % orig_domain;
    for (int g = 0; g < n_groups; g++) merkle_prove(&groups[g]->merkle, io, pos);
    for (int r = 0; r < n_rounds; r++) {
      size_t group = pos % (rounds[r].domain / ORC_FRI_FOLD);
      merkle_prove(&rounds[r].merkle, io, group);
      pos = group;
    }
  }
  for (int r = 0; r < n_rounds; r++) { free(rounds[r].evaluated); free(rounds[r].merkle.nodes); }
}

/* ------------------------------------------------------------------ the sequencer */
size_t orc_prove_segment(const orc_circuit_t* c, const uint32_t* blob, size_t blob_words, uint32_t po2,
                         const uint32_t* code, const uint32_t* data, const uint32_t* global, uint32_t* seal,
                         size_t seal_cap) {
  const size_t n = (size_t)1 << po2;
  wiop_t io;
  memset(&io, 0, sizeof io);
  orc_rng_init(&io.rng);
  (void)blob; (void)blob_words;
  transcript_seed(&io.rng, c);

  /* globals ++ po2: the seal opens with all of them; the transcript takes the early ones (and po2) here, the late ones -- public
   * inputs that depend on commitments made outside this proof -- after the DATA commitment */
  {
    uint32_t ng = c->n_global, ne = ng - c->n_late, d[8];
    uint32_t* v = (uint32_t*)malloc(4 * (ng + 1));
    memcpy(v, global, 4 * ne);
    v[ne] = fp_enc(po2);
    orc_hash_elem_slice(v, ne + 1, d);
    orc_rng_mix(&io.rng, d);
    memcpy(v, global, 4 * ng);
    v[ng] = fp_enc(po2);
    wiop_write(&io, v, ng + 1);
    free(v);
  }

  group_t grp[3], check;
  group_from_witness(&grp[ORC_GROUP_CODE], code, c->group_size[ORC_GROUP_CODE], po2);
  merkle_commit(&grp[ORC_GROUP_CODE].merkle, &io);
  group_from_witness(&grp[ORC_GROUP_DATA], data, c->group_size[ORC_GROUP_DATA], po2);
  merkle_commit(&grp[ORC_GROUP_DATA].merkle, &io);
  if (c->n_late) {
    uint32_t d[8];
    orc_hash_elem_slice(global + (c->n_global - c->n_late), c->n_late, d);
    orc_rng_mix(&io.rng, d);
  }

  uint32_t* mix = (uint32_t*)malloc(4 * (c->n_mix ? c->n_mix : 1));
  for (uint32_t i = 0; i < c->n_mix; i++) mix[i] = orc_rng_elem(&io.rng);
  {
    uint32_t* accum = (uint32_t*)malloc((size_t)c->group_size[ORC_GROUP_ACCUM] * n * 4);
    orc_accum_public(c, po2, code, data, global, mix, accum);
    group_from_witness(&grp[ORC_GROUP_ACCUM], accum, c->group_size[ORC_GROUP_ACCUM], po2);
    free(accum);
  }
  merkle_commit(&grp[ORC_GROUP_ACCUM].merkle, &io);

  /* constraint (check) polynomial */
  fp4_t poly_mix = rng_ext(&io.rng);
  check.count = ORC_CHECK_SIZE;
  check.coeffs = (uint32_t*)malloc(n * ORC_INV_RATE * 16);
  orc_eval_check(c, po2, grp[0].evaluated, grp[1].evaluated, grp[2].evaluated, global, mix, poly_mix.e, check.coeffs);
  orc_batch_interpolate_ntt(check.coeffs, 4, po2 + 2); /* 4 polys of 4N == 16 polys of N, bit-reversed */
  group_finish(&check, po2);
  merkle_commit(&check.merkle, &io);

  /* DEEP point and the tap evaluations */
  fp4_t z = rng_ext(&io.rng);
  fp_t back_one = orc_rou_rev(po2);
  uint32_t n_u = c->n_taps + ORC_CHECK_SIZE;
  uint32_t* all_xs = (uint32_t*)malloc(16 * (size_t)n_u);
  uint32_t* eval_u = (uint32_t*)malloc(16 * (size_t)n_u);
  uint32_t* coeff_u = (uint32_t*)malloc(16 * (size_t)n_u);
  uint32_t* which = (uint32_t*)malloc(4 * (size_t)n_u);
  for (uint32_t t = 0; t < c->n_taps; t++) {
    st4(all_xs + 4 * t, fp4_scale(z, fp_pow(back_one, c->taps[t].back)));
    which[t] = c->taps[t].offset;
  }
  for (int g = 0; g < 3; g++) {
    uint32_t b = c->group_tap_begin[g], e = c->group_tap_begin[g + 1];
    orc_batch_evaluate_any(grp[g].coeffs, po2, which + b, all_xs + 4 * b, e - b, eval_u + 4 * b);
  }
  for (uint32_t r = 0; r < c->n_regs; r++) {
    uint32_t p = c->regs[r].first_tap;
    orc_poly_interpolate(coeff_u + 4 * p, all_xs + 4 * p, eval_u + 4 * p, c->regs[r].size);
  }
  fp4_t z4 = fp4_pow(z, 4);
  for (uint32_t i = 0; i < ORC_CHECK_SIZE; i++) { which[c->n_taps + i] = i; st4(all_xs + 4 * (c->n_taps + i), z4); }
  orc_batch_evaluate_any(check.coeffs, po2, which + c->n_taps, all_xs + 4 * c->n_taps, ORC_CHECK_SIZE, coeff_u + 4 * c->n_taps);
  wiop_write(&io, coeff_u, 4 * (size_t)n_u);
  {
    uint32_t d[8];
    orc_hash_elem_slice(coeff_u, 4 * (size_t)n_u, d);
    orc_rng_mix(&io.rng, d);
  }

  /* FRI batching */
  fp4_t mixv = rng_ext(&io.rng);
  uint32_t n_combos = c->n_combos;
  uint32_t* combos = (uint32_t*)calloc((size_t)(n_combos + 1) * n, 16);
  fp4_t cur = fp4_one();
  for (int g = 0; g < 3; g++) {
    uint32_t gs = c->group_size[g];
    uint32_t* combo_of = (uint32_t*)malloc(4 * (gs ? gs : 1));
    for (uint32_t r = 0; r < c->n_regs; r++)
      if (c->regs[r].group == (uint32_t)g) combo_of[c->regs[r].offset] = c->regs[r].combo;
    orc_mix_poly_coeffs(combos, cur.e, mixv.e, grp[g].coeffs, combo_of, gs, po2);
    cur = fp4_mul(cur, fp4_pow(mixv, gs));
    free(combo_of);
  }
  {
    uint32_t combo_of[ORC_CHECK_SIZE];
    for (int i = 0; i < ORC_CHECK_SIZE; i++) combo_of[i] = n_combos;
    orc_mix_poly_coeffs(combos, cur.e, mixv.e, check.coeffs, combo_of, ORC_CHECK_SIZE, po2);
  }
  /* subtract the interpolants, divide out the DEEP denominators */
  cur = fp4_one();
  for (uint32_t r = 0; r < c->n_regs; r++) {
    for (uint32_t i = 0; i < c->regs[r].size; i++) {
      uint32_t* dst = combos + ((size_t)c->regs[r].combo * n + i) * 4;
      st4(dst, fp4_sub(ld4(dst), fp4_mul(cur, ld4(coeff_u + 4 * (c->regs[r].first_tap + i)))));
    }
    cur = fp4_mul(cur, mixv);
  }
  for (uint32_t i = 0; i < ORC_CHECK_SIZE; i++) {
    uint32_t* dst = combos + (size_t)n_combos * n * 4;
    st4(dst, fp4_sub(ld4(dst), fp4_mul(cur, ld4(coeff_u + 4 * (c->n_taps + i)))));
    cur = fp4_mul(cur, mixv);
  }
  int ok = 1;
  for (uint32_t k = 0; k < n_combos; k++)
    for (uint32_t b = c->combo_begin[k]; b < c->combo_begin[k + 1]; b++) {
      fp4_t pt = fp4_scale(z, fp_pow(back_one, c->combo_backs[b])), rem;
      orc_poly_divide(combos + (size_t)k * n * 4, (uint32_t)n, pt.e, rem.e);
      ok &= fp4_eq(rem, fp4_zero());
    }
  {
    fp4_t rem;
    orc_poly_divide(combos + (size_t)n_combos * n * 4, (uint32_t)n, z4.e, rem.e);
    ok &= fp4_eq(rem, fp4_zero());
  }
  uint32_t* final_coeffs = (uint32_t*)malloc(n * 16);
  orc_eltwise_sum_extelem(final_coeffs, combos, n_combos + 1, (uint32_t)n);
  orc_batch_bit_reverse(final_coeffs, 4, po2);
  free(combos); free(all_xs); free(eval_u); free(coeff_u); free(which); free(mix);

  group_t* order[4] = {&grp[0], &grp[1], &grp[2], &check};
  fri_prove(&io, final_coeffs, n, order, 4);
  for (int g = 0; g < 3; g++) group_free(&grp[g]);
  group_free(&check);

  size_t len = ok ? io.n : 0;
  if (seal && len && len <= seal_cap) memcpy(seal, io.w, len * 4);
  free(io.w);
  return len;
}

/* ================================================================== verifier */
typedef struct { const uint32_t* w; size_t n, pos; orc_rng_t rng; int underflow, bad_elem; } riop_t;
static const uint32_t* riop_read(riop_t* io, size_t n) {
  static const uint32_t zeros[4096] = {0};
  if (io->pos + n > io->n) { io->underflow = 1; return n <= 4096 ? zeros : NULL; }
  const uint32_t* p = io->w + io->pos;
  io->pos += n;
  return p;
}

typedef struct { orc_merkle_params_t p; uint32_t* top; } vmerkle_t;
static int vmerkle_new(vmerkle_t* m, riop_t* io, size_t rows, size_t cols) {
  orc_merkle_params(&m->p, rows, cols, ORC_QUERIES);
  size_t ts = m->p.top_size;
  m->top = (uint32_t*)calloc(ts * 2, 32);
  const uint32_t* src = riop_read(io, ts * 8);
  if (io->underflow || !src) return 1;
  for (size_t i = 0; i < ts * 8; i++) if (src[i] >= ORC_P) return 2; /* digest words are field elements */
  memcpy(m->top + ts * 8, src, ts * 32);
  for (size_t i = ts; i-- > 1;) orc_hash_pair(m->top + 2 * i * 8, m->top + (2 * i + 1) * 8, m->top + i * 8);
  orc_rng_mix(&io->rng, m->top + 8);
  return 0;
}
/* returns pointer to the column values (col_size words) or NULL on failure */
static const uint32_t* vmerkle_verify(vmerkle_t* m, riop_t* io, size_t idx) {
  if (idx >= m->p.row_size) return NULL;
  const uint32_t* col = riop_read(io, m->p.col_size);
  if (io->underflow || !col) return NULL;
  for (size_t i = 0; i < m->p.col_size; i++) if (col[i] >= ORC_P) return NULL;
  uint32_t cur[8], nxt[8];
  orc_hash_elem_slice(col, m->p.col_size, cur);
  idx += m->p.row_size;
  while (idx >= 2 * m->p.top_size) {
    const uint32_t* other = riop_read(io, 8);
    if (io->underflow) return NULL;
    for (int i = 0; i < 8; i++) if (other[i] >= ORC_P) { io->bad_elem = 1; return NULL; }
    if (idx & 1) orc_hash_pair(other, cur, nxt); else orc_hash_pair(cur, other, nxt);
    memcpy(cur, nxt, 32);
    idx /= 2;
  }
  return memcmp(m->top + idx * 8, cur, 32) == 0 ? col : NULL;
}

static fp4_t poly_eval4(const fp4_t* coeffs, size_t n, fp4_t x) {
  fp4_t tot = fp4_zero();
  for (size_t i = n; i-- > 0;) tot = fp4_add(fp4_mul(tot, x), coeffs[i]);
  return tot;
}

enum {
  V_OK = 0, V_TRUNCATED, V_BAD_PO2, V_MERKLE_GROUP, V_CHECK_MISMATCH, V_FRI_MERKLE, V_FRI_GOAL, V_FRI_FINAL, V_TRAILING,
  V_BAD_ELEM, V_CODE_ROOT
};
const char* orc_verify_strerror(int code) {
  static const char* const names[] = {"ok", "seal truncated", "bad po2", "group merkle path rejected",
                                      "constraint check mismatch at z", "fri merkle path rejected", "fri fold goal mismatch",
                                      "fri final polynomial mismatch", "trailing words in seal", "non-canonical field element",
                                      "code root is not the expected control root"};
  return code >= 0 && code <= V_CODE_ROOT ? names[code] : "unknown";
}

int orc_verify_segment(const orc_circuit_t* c, const uint32_t* blob, size_t blob_words, const uint32_t* seal,
                       size_t seal_words) {
  return orc_verify_segment_bound(c, blob, blob_words, seal, seal_words, NULL);
}

/* risc0-zkp verify/mod.rs: `check_code(po2, &code_root)` -- the CODE commitment must be the program's control root */
int orc_verify_segment_bound(const orc_circuit_t* c, const uint32_t* blob, size_t blob_words, const uint32_t* seal,
                             size_t seal_words, const uint32_t* expected_code_root) {
  riop_t io;
  memset(&io, 0, sizeof io);
  io.w = seal; io.n = seal_words;
  orc_rng_init(&io.rng);
  (void)blob; (void)blob_words;
  transcript_seed(&io.rng, c);
  int rc = V_OK;

  uint32_t ng = c->n_global;
  const uint32_t* gvec = riop_read(&io, ng + 1);
  if (io.underflow) return V_TRUNCATED;
  for (uint32_t i = 0; i <= ng; i++) if (gvec[i] >= ORC_P) return V_BAD_ELEM;
  uint32_t po2 = fp_dec(gvec[ng]);
  if (po2 < 9 || po2 > 24) return V_BAD_PO2;
  {
    uint32_t d[8], ne = ng - c->n_late;
    uint32_t* early = (uint32_t*)malloc(4 * (ne + 1));
    memcpy(early, gvec, 4 * ne);
    early[ne] = gvec[ng];
    orc_hash_elem_slice(early, ne + 1, d);
    orc_rng_mix(&io.rng, d);
    free(early);
  }
  const size_t n = (size_t)1 << po2, domain = n * ORC_INV_RATE;

  vmerkle_t vm[4];
  memset(vm, 0, sizeof vm);
  uint32_t* mix = (uint32_t*)malloc(4 * (c->n_mix ? c->n_mix : 1));
  uint32_t n_u = c->n_taps + ORC_CHECK_SIZE;
  fp4_t* coeff_u = (fp4_t*)malloc(sizeof(fp4_t) * n_u);
  fp4_t* eval_u = (fp4_t*)malloc(sizeof(fp4_t) * n_u);
  uint32_t max_combo = 1;
  for (uint32_t k = 0; k < c->n_combos; k++) {
    uint32_t sz = c->combo_begin[k + 1] - c->combo_begin[k];
    if (sz > max_combo) max_combo = sz;
  }
  fp4_t* combo_u = (fp4_t*)calloc((size_t)(c->n_combos + 1) * max_combo, sizeof(fp4_t));
  fp4_t* combo_tot = (fp4_t*)malloc(sizeof(fp4_t) * (c->n_combos + 1));
  vmerkle_t fm[8];
  fp4_t fmix[8];
  size_t fdomain[8];
  int n_rounds = 0;
  fp4_t* final_poly = NULL;

  int mr;
#define VM_NEW(m, rows, cols) do { if ((mr = vmerkle_new(m, &io, rows, cols)) != 0) { rc = mr == 2 ? V_BAD_ELEM : V_TRUNCATED; goto done; } } while (0)
  VM_NEW(&vm[ORC_GROUP_CODE], domain, c->group_size[ORC_GROUP_CODE]);
  if (expected_code_root && memcmp(expected_code_root, vm[ORC_GROUP_CODE].top + 8, 32) != 0) { rc = V_CODE_ROOT; goto done; }
  VM_NEW(&vm[ORC_GROUP_DATA], domain, c->group_size[ORC_GROUP_DATA]);
  if (c->n_late) {
    uint32_t d[8];
    orc_hash_elem_slice(gvec + (ng - c->n_late), c->n_late, d);
    orc_rng_mix(&io.rng, d);
  }
  for (uint32_t i = 0; i < c->n_mix; i++) mix[i] = orc_rng_elem(&io.rng);
  VM_NEW(&vm[ORC_GROUP_ACCUM], domain, c->group_size[ORC_GROUP_ACCUM]);
  fp4_t poly_mix = rng_ext(&io.rng);
  VM_NEW(&vm[3], domain, ORC_CHECK_SIZE);
  fp4_t z = rng_ext(&io.rng);
  fp_t back_one = orc_rou_rev(po2);
  {
    const uint32_t* cu = riop_read(&io, 4 * (size_t)n_u);
    if (io.underflow || !cu) { rc = V_TRUNCATED; goto done; }
    for (size_t i = 0; i < 4 * (size_t)n_u; i++) if (cu[i] >= ORC_P) { rc = V_BAD_ELEM; goto done; }
    memcpy(coeff_u, cu, 16 * (size_t)n_u);
    uint32_t d[8];
    orc_hash_elem_slice(cu, 4 * (size_t)n_u, d);
    orc_rng_mix(&io.rng, d);
  }
  /* tap values at z from the interpolants, then the constraint identity */
  for (uint32_t r = 0; r < c->n_regs; r++) {
    uint32_t p = c->regs[r].first_tap, sz = c->regs[r].size;
    for (uint32_t i = 0; i < sz; i++)
      eval_u[p + i] = poly_eval4(coeff_u + p, sz, fp4_scale(z, fp_pow(back_one, c->taps[p + i].back)));
  }
  {
    fp4_t result;
    orc_poly_ext(c, poly_mix.e, (const uint32_t*)eval_u, gvec, mix, result.e);
    fp4_t check = fp4_zero();
    for (uint32_t r = 0; r < 4; r++) {
      fp4_t zr = fp4_pow(z, r);
      uint32_t blk = orc_bitrev(r, 2);
      for (uint32_t j = 0; j < 4; j++) {
        fp4_t basis = fp4_zero();
        basis.e[j] = ORC_ONE;
        check = fp4_add(check, fp4_mul(fp4_mul(coeff_u[c->n_taps + 4 * j + blk], zr), basis));
      }
    }
    fp4_t van = fp4_sub(fp4_pow(fp4_scale(z, fp_enc(3)), n), fp4_one());
    check = fp4_mul(check, van);
    if (!fp4_eq(check, result)) { rc = V_CHECK_MISMATCH; goto done; }
  }
  /* FRI batching combination of the interpolants */
  fp4_t mixv = rng_ext(&io.rng);
  {
    fp4_t cur = fp4_one();
    for (uint32_t r = 0; r < c->n_regs; r++) {
      for (uint32_t i = 0; i < c->regs[r].size; i++) {
        fp4_t* dst = &combo_u[(size_t)c->regs[r].combo * max_combo + i];
        *dst = fp4_add(*dst, fp4_mul(cur, coeff_u[c->regs[r].first_tap + i]));
      }
      cur = fp4_mul(cur, mixv);
    }
    for (uint32_t i = 0; i < ORC_CHECK_SIZE; i++) {
      fp4_t* dst = &combo_u[(size_t)c->n_combos * max_combo];
      *dst = fp4_add(*dst, fp4_mul(cur, coeff_u[c->n_taps + i]));
      cur = fp4_mul(cur, mixv);
    }
  }
  /* FRI commitments */
  size_t degree = n, dom = domain;
  while (degree > ORC_FRI_MIN_DEGREE) {
    fdomain[n_rounds] = dom / ORC_FRI_FOLD;
    memset(&fm[n_rounds], 0, sizeof(vmerkle_t));
    if ((mr = vmerkle_new(&fm[n_rounds], &io, dom / ORC_FRI_FOLD, ORC_FRI_FOLD * 4)) != 0) { n_rounds++; rc = mr == 2 ? V_BAD_ELEM : V_TRUNCATED; goto done; }
    fmix[n_rounds] = rng_ext(&io.rng);
    n_rounds++;
    dom /= ORC_FRI_FOLD;
    degree /= ORC_FRI_FOLD;
  }
  {
    const uint32_t* fc = riop_read(&io, 4 * degree);
    if (io.underflow || !fc) { rc = V_TRUNCATED; goto done; }
    for (size_t i = 0; i < 4 * degree; i++) if (fc[i] >= ORC_P) { rc = V_BAD_ELEM; goto done; }
    uint32_t d[8];
    orc_hash_elem_slice(fc, 4 * degree, d);
    orc_rng_mix(&io.rng, d);
    final_poly = (fp4_t*)malloc(sizeof(fp4_t) * degree);
    for (size_t i = 0; i < degree; i++)
      for (int k = 0; k < 4; k++) final_poly[i].e[k] = fc[(size_t)k * degree + i];
  }
  const fp_t gen_final = orc_rou_fwd(log2u(dom));
  const fp_t gen_domain = orc_rou_fwd(log2u(domain));
  for (int q = 0; q < ORC_QUERIES; q++) {
    size_t pos = orc_rng_bits(&io.rng, log2u(domain)) % domain;
    /* DEEP quotient at x = w^pos from the opened rows */
    const uint32_t* rows[4];
    for (int g = 0; g < 4; g++) {
      rows[g] = vmerkle_verify(&vm[g], &io, pos);
      if (!rows[g]) { rc = io.underflow ? V_TRUNCATED : io.bad_elem ? V_BAD_ELEM : V_MERKLE_GROUP; goto done; }
    }
    fp_t x = fp_pow(gen_domain, pos);
    fp4_t cur = fp4_one();
    for (uint32_t k = 0; k <= c->n_combos; k++) combo_tot[k] = fp4_zero();
    for (uint32_t r = 0; r < c->n_regs; r++) {
      combo_tot[c->regs[r].combo] = fp4_add(combo_tot[c->regs[r].combo], fp4_scale(cur, rows[c->regs[r].group][c->regs[r].offset]));
      cur = fp4_mul(cur, mixv);
    }
    for (uint32_t i = 0; i < ORC_CHECK_SIZE; i++) {
      combo_tot[c->n_combos] = fp4_add(combo_tot[c->n_combos], fp4_scale(cur, rows[3][i]));
      cur = fp4_mul(cur, mixv);
    }
    fp4_t goal = fp4_zero(), xe = fp4_from_fp(x);
    for (uint32_t k = 0; k < c->n_combos; k++) {
      uint32_t sz = c->combo_begin[k + 1] - c->combo_begin[k];
      fp4_t num = fp4_sub(combo_tot[k], poly_eval4(combo_u + (size_t)k * max_combo, sz, xe));
      fp4_t den = fp4_one();
      for (uint32_t b = c->combo_begin[k]; b < c->combo_begin[k + 1]; b++)
        den = fp4_mul(den, fp4_sub(xe, fp4_scale(z, fp_pow(back_one, c->combo_backs[b]))));
      goal = fp4_add(goal, fp4_mul(num, fp4_inv(den)));
    }
    {
      fp4_t num = fp4_sub(combo_tot[c->n_combos], combo_u[(size_t)c->n_combos * max_combo]);
      goal = fp4_add(goal, fp4_mul(num, fp4_inv(fp4_sub(xe, fp4_pow(z, 4)))));
    }
    /* fold rounds */
    for (int r = 0; r < n_rounds; r++) {
      size_t rd = fdomain[r], quot = pos / rd, group = pos % rd;
      const uint32_t* col = vmerkle_verify(&fm[r], &io, group);
      if (!col) { rc = io.underflow ? V_TRUNCATED : io.bad_elem ? V_BAD_ELEM : V_FRI_MERKLE; goto done; }
      fp4_t v[ORC_FRI_FOLD];
      for (int i = 0; i < ORC_FRI_FOLD; i++)
        for (int k = 0; k < 4; k++) v[i].e[k] = col[k * ORC_FRI_FOLD + i];
      if (!fp4_eq(v[quot], goal)) { rc = V_FRI_GOAL; goto done; }
      /* size-16 inverse DFT over the coset {w^group * zeta^k}, un-twist by w^-group, combine with the fold mix */
      fp_t zeta_inv = orc_rou_rev(4), inv16 = fp_inv(fp_enc(16));
      fp_t inv_wk = fp_pow(orc_rou_rev(log2u(rd * ORC_FRI_FOLD)), group);
      fp4_t tot = fp4_zero(), mixpow = fp4_one();
      fp_t twist = ORC_ONE;
      for (int j = 0; j < ORC_FRI_FOLD; j++) {
        fp4_t cj = fp4_zero();
        for (int k = 0; k < ORC_FRI_FOLD; k++) cj = fp4_add(cj, fp4_scale(v[k], fp_pow(zeta_inv, (uint64_t)j * k)));
        cj = fp4_scale(cj, fp_mul(inv16, twist));
        tot = fp4_add(tot, fp4_mul(cj, mixpow));
        mixpow = fp4_mul(mixpow, fmix[r]);
        twist = fp_mul(twist, inv_wk);
      }
      goal = tot;
      pos = group;
    }
    fp4_t fx = poly_eval4(final_poly, degree, fp4_from_fp(fp_pow(gen_final, pos)));
    if (!fp4_eq(fx, goal)) { rc = V_FRI_FINAL; goto done; }
  }
  if (io.pos != io.n) rc = V_TRAILING;
done:
  for (int g = 0; g < 4; g++) free(vm[g].top);
  for (int r = 0; r < n_rounds; r++) free(fm[r].top);
  free(mix); free(coeff_u); free(eval_u); free(combo_u); free(combo_tot); free(final_poly);
  return rc;
}
