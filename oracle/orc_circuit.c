/* ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_field.h).  Circuit blob, witness generation, accumulation and
 * constraint evaluation on the CPU.
 *
 * Upstream modules restated (un-vendored, recalled):
 *   taps / registers / combos   risc0-zkp taps.rs (`TapSet`, `RegisterRef`, combos = distinct back-sets)
 *   constraint program          risc0-zkp adapter.rs (`PolyExtStep::{Const,Get,GetGlobal,Add,Sub,Mul,True,AndEqz,AndCond}`,
 *                               `MixState{tot,mul}`)
 *   eval_check                  risc0-circuit-rv32im-sys 4.0.2 kernels `eval_check` (+ generated `poly_fp`):
 *                               check[i] = poly(taps at i - 4*back) / ((3 w^i)^N - 1) on the 4N domain
 *   witgen / accum              risc0-circuit-rv32im 4.0.4 `generate_witness`, `step_accum` + hal `prefix_products`;
 *                               the real rv32im step functions are not reproducible here, so the blob carries a
 *                               *synthetic* column program of the same shape (free/derived data columns, grand-product
 *                               accumulators gated by a CODE "first row" selector).
 */
#include "orc_circuit.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define BLOB_MAGIC 0x31433052u
enum { SEC_GROUPS = 1, SEC_TAPS = 2, SEC_GLOBALS = 3, SEC_POLY = 4, SEC_WITGEN = 5, SEC_ACCUM = 6, SEC_INFO = 7, SEC_ACCUM_FP = 8, SEC_LATE = 9, SEC_LOGUP = 10, SEC_PERIODIC = 11, SEC_SPONGE = 12 };
#define TAG_AND (1u << 24)

uint32_t orc_circuit_group_size(const orc_circuit_t* c, uint32_t g) { return c->group_size[g]; }
uint32_t orc_circuit_n_taps(const orc_circuit_t* c) { return c->n_taps; }
uint32_t orc_circuit_n_global(const orc_circuit_t* c) { return c->n_global; }
uint32_t orc_circuit_n_mix(const orc_circuit_t* c) { return c->n_mix; }
uint32_t orc_circuit_n_late(const orc_circuit_t* c) { return c->n_late; }
uint32_t orc_circuit_n_combos(const orc_circuit_t* c) { return c->n_combos; }

void orc_circuit_free(orc_circuit_t* c) {
  if (!c) return;
  free(c->taps); free(c->regs); free(c->combo_begin); free(c->combo_backs); free(c->steps);
  free(c->code_cols); free(c->data_cols); free(c->acc_cols); free(c->acc_fp); free(c->global_cols); free(c->logup); free(c->logup_words); free(c->periodic); free(c);
}

static void derive_regs_and_combos(orc_circuit_t* c) {
  /* registers: maximal runs of taps sharing (group, offset); combos: distinct back-lists, numbered by first appearance */
  c->regs = (orc_reg_t*)calloc(c->n_taps ? c->n_taps : 1, sizeof(orc_reg_t));
  c->combo_begin = (uint32_t*)calloc(c->n_taps + 2, sizeof(uint32_t));
  c->combo_backs = (uint32_t*)calloc(c->n_taps + 1, sizeof(uint32_t));
  c->n_regs = 0; c->n_combos = 0;
  for (int g = 0; g < 4; g++) c->group_tap_begin[g] = c->n_taps;
  for (uint32_t t = 0; t < c->n_taps;) {
    uint32_t e = t;
    while (e < c->n_taps && c->taps[e].group == c->taps[t].group && c->taps[e].offset == c->taps[t].offset) e++;
    uint32_t size = e - t, combo = c->n_combos;
    for (uint32_t k = 0; k < c->n_combos; k++) {
      uint32_t b = c->combo_begin[k], len = c->combo_begin[k + 1] - b;
      if (len != size) continue;
      int same = 1;
      for (uint32_t i = 0; i < size; i++) same &= c->combo_backs[b + i] == c->taps[t + i].back;
      if (same) { combo = k; break; }
    }
    if (combo == c->n_combos) {
      uint32_t b = c->combo_begin[c->n_combos];
      for (uint32_t i = 0; i < size; i++) c->combo_backs[b + i] = c->taps[t + i].back;
      c->combo_begin[++c->n_combos] = b + size;
    }
    orc_reg_t* r = &c->regs[c->n_regs++];
    r->group = c->taps[t].group; r->offset = c->taps[t].offset; r->first_tap = t; r->size = size; r->combo = combo;
    t = e;
  }
  for (uint32_t t = c->n_taps; t-- > 0;) c->group_tap_begin[c->taps[t].group] = t;
  for (int g = 2; g >= 0; g--)
    if (c->group_tap_begin[g] == c->n_taps && g < 3) c->group_tap_begin[g] = c->group_tap_begin[g + 1];
}

orc_circuit_t* orc_circuit_parse(const uint32_t* w, size_t n_words) {
  if (n_words < 3 || w[0] != BLOB_MAGIC || w[1] != 1) return NULL;
  orc_circuit_t* c = (orc_circuit_t*)calloc(1, sizeof *c);
  memcpy(c->info, "R0HIP_SYNTH:v1__", 16);
  size_t pos = 3;
  for (uint32_t s = 0; s < w[2]; s++) {
    if (pos + 2 > n_words) goto bad;
    uint32_t tag = w[pos], len = w[pos + 1];
    const uint32_t* p = w + pos + 2;
    if (pos + 2 + len > n_words) goto bad;
    switch (tag) {
      case SEC_GROUPS: memcpy(c->group_size, p, 12); break;
      case SEC_TAPS:
        c->n_taps = p[0];
        c->taps = (orc_tap_t*)malloc(sizeof(orc_tap_t) * (c->n_taps ? c->n_taps : 1));
        memcpy(c->taps, p + 1, sizeof(orc_tap_t) * c->n_taps);
        break;
      case SEC_GLOBALS:
        c->n_global = p[0]; c->n_mix = p[1];
        c->global_cols = (uint32_t*)calloc(c->n_global ? c->n_global : 1, 4);
        if (len >= 2 + c->n_global) memcpy(c->global_cols, p + 2, 4 * c->n_global);
        break;
      case SEC_POLY:
        c->n_steps = p[0]; c->ret = p[1];
        c->steps = (orc_step_t*)malloc(sizeof(orc_step_t) * (c->n_steps ? c->n_steps : 1));
        memcpy(c->steps, p + 2, sizeof(orc_step_t) * c->n_steps);
        break;
      case SEC_WITGEN: {
        c->n_code = p[0];
        c->code_cols = (orc_code_col_t*)malloc(sizeof(orc_code_col_t) * (c->n_code ? c->n_code : 1));
        memcpy(c->code_cols, p + 1, sizeof(orc_code_col_t) * c->n_code);
        const uint32_t* q = p + 1 + 2 * c->n_code;
        c->n_data = q[0];
        c->data_cols = (orc_data_col_t*)malloc(sizeof(orc_data_col_t) * (c->n_data ? c->n_data : 1));
        memcpy(c->data_cols, q + 1, sizeof(orc_data_col_t) * c->n_data);
        break;
      }
      case SEC_ACCUM:
        c->n_acc = p[0];
        c->acc_cols = (orc_acc_col_t*)malloc(sizeof(orc_acc_col_t) * (c->n_acc ? c->n_acc : 1));
        memcpy(c->acc_cols, p + 1, sizeof(orc_acc_col_t) * c->n_acc);
        break;
      case SEC_ACCUM_FP:
        if (len < 1 || len != 1 + 13 * (size_t)p[0]) goto bad;
        c->n_acc_fp = p[0];
        c->acc_fp = (orc_acc_fp_t*)malloc(sizeof(orc_acc_fp_t) * (c->n_acc_fp ? c->n_acc_fp : 1));
        memcpy(c->acc_fp, p + 1, sizeof(orc_acc_fp_t) * c->n_acc_fp);
        break;
      case SEC_INFO:
        if (len >= 4) memcpy(c->info, p, 16);
        break;
      case SEC_LATE:
        if (len != 1) goto bad;
        c->n_late = p[0];
        break;
      case SEC_PERIODIC:
        if (len < 2 || !p[0] || (uint64_t)p[0] * p[1] + 2 != len || c->periodic) goto bad;
        c->period = p[0]; c->n_periodic = p[1];
        c->periodic = (uint32_t*)malloc(sizeof(uint32_t) * (len - 1));
        memcpy(c->periodic, p + 2, sizeof(uint32_t) * (len - 2));
        for (uint32_t k = 0; k + 2 < len; k++) if (c->periodic[k] >= ORC_P) goto bad;
        break;
      case SEC_SPONGE:
        if (len != 3) goto bad;
        c->has_sponge = 1; c->sponge_code = p[0]; c->sponge_data = p[1]; c->sponge_global = p[2];
        break;
      case SEC_LOGUP: {
        if (len < 2 || c->logup_words) goto bad;
        c->logup_words = (uint32_t*)malloc(4 * (size_t)len);
        memcpy(c->logup_words, p, 4 * (size_t)len);
        const uint32_t* q = c->logup_words;
        size_t at = 0;
#define WORD(dst) do { if (at >= len) goto bad; (dst) = q[at++]; } while (0)
#define FORM(lf) do { WORD((lf).n); if ((lf).n > 64 || at + 3 * (size_t)(lf).n > len) goto bad; (lf).t = (const orc_lf_term_t*)(q + at); at += 3 * (size_t)(lf).n; } while (0)
        WORD(c->n_logup); WORD(c->n_tables);
        if (c->n_logup < 1 || c->n_logup > 64 || c->n_tables > 8) goto bad;
        for (uint32_t k = 0; k < c->n_tables; k++) { WORD(c->table_col[k]); WORD(c->table_kind[k]); }
        c->logup = (orc_logup_acc_t*)calloc(c->n_logup, sizeof(orc_logup_acc_t));
        int chain_over = 0;
        for (uint32_t j = 0; j < c->n_logup; j++) {
          orc_logup_acc_t* a = &c->logup[j];
          uint32_t nf;
          WORD(nf); WORD(a->final_global);
          if (nf != 4) goto bad;
          if (a->final_global == 0xffffffffu) { if (chain_over) goto bad; c->n_chain++; } else chain_over = 1;
          for (uint32_t f = 0; f < 4; f++) {
            orc_logup_fraction_t* fr = &a->fr[f];
            WORD(fr->table);
            FORM(fr->num);
            WORD(fr->n_parts);
            if (fr->table > 2 || fr->n_parts < 1 || fr->n_parts > 8) goto bad;
            for (uint32_t k = 0; k < fr->n_parts; k++) { WORD(fr->parts[k].ch_kind); WORD(fr->parts[k].ch_idx); FORM(fr->parts[k].lf); if (fr->parts[k].ch_kind > 2) goto bad; }
            if (fr->table && (fr->n_parts != 2 || fr->parts[1].ch_kind != 0)) goto bad;
          }
        }
        if (at != len) goto bad;
#undef WORD
#undef FORM
        break;
      }
      default: break;
    }
    pos += 2 + len;
  }
  for (uint32_t i = 0; i < c->n_steps; i++) {
    uint32_t op = c->steps[i].op;
    if (op == OP_TRUE || op == OP_AND_EQZ || op == OP_AND_COND) c->n_mix_vars++;
    else c->n_fp_vars++;
  }
  /* WITGEN/ACCUM (the synthetic column program) are optional: circuits imported from risc0 tables omit both */
  if ((c->code_cols || c->acc_cols || c->acc_fp || c->logup) && (c->n_code != c->group_size[ORC_GROUP_CODE] || c->n_data != c->group_size[ORC_GROUP_DATA])) goto bad;
  if (c->n_late > c->n_global) goto bad;
  for (uint32_t k = 0; c->code_cols && k < c->n_code; k++)
    if (c->code_cols[k].kind == 6 && (!c->periodic || c->code_cols[k].param >= c->n_periodic)) goto bad;
  if (c->has_sponge && ((uint64_t)c->sponge_code + 28 > c->n_code || (uint64_t)c->sponge_data + 65 > c->n_data || (uint64_t)c->sponge_global + 8 > c->n_global || c->period != 30)) goto bad;
  if (c->logup) {  /* every reference of the log-derivative argument stays inside the circuit */
    if (c->acc_cols || c->acc_fp || 4 * c->n_logup != c->group_size[ORC_GROUP_ACCUM] || !c->n_chain) goto bad;
    for (uint32_t k = 0; k < c->n_tables; k++)
      if (c->table_col[k] >= c->n_data || (c->table_kind[k] != 1 && c->table_kind[k] != 2)) goto bad;
    for (uint32_t j = 0; j < c->n_logup; j++) {
      const orc_logup_acc_t* a = &c->logup[j];
      if (a->final_global != 0xffffffffu && (uint64_t)a->final_global + 4 > c->n_global) goto bad;
      for (int f = 0; f < 4; f++) {
        const orc_logup_fraction_t* fr = &a->fr[f];
        for (uint32_t k = 0; k <= fr->n_parts; k++) {
          const orc_lf_t* lf = k ? &fr->parts[k - 1].lf : &fr->num;
          for (uint32_t t = 0; t < lf->n; t++) {
            if (lf->t[t].coef >= ORC_P || lf->t[t].global > c->n_global) goto bad;
            if (lf->t[t].col) {
              uint32_t ref = lf->t[t].col - 1, g = ref >> 28, col = ref & 0xfffffu;
              if ((g != ORC_GROUP_CODE && g != ORC_GROUP_DATA) || col >= c->group_size[g]) goto bad;
            }
          }
          if (k) {
            const orc_logup_part_t* q = &fr->parts[k - 1];
            if ((q->ch_kind == 1 && 4 * (uint64_t)q->ch_idx + 4 > c->n_mix) || (q->ch_kind == 2 && (uint64_t)q->ch_idx + 4 > c->n_global)) goto bad;
          }
        }
      }
    }
  } else
  if (c->acc_fp) {  /* the trace circuit's memory-consistency accumulators: alpha, b1, b2, b3 shared */
    if (c->acc_cols || 4 * c->n_acc_fp != c->group_size[ORC_GROUP_ACCUM] || c->n_mix != 16) goto bad;
    for (uint32_t j = 0; j < c->n_acc_fp; j++) {
      if (c->acc_fp[j].n_f < 1 || c->acc_fp[j].n_f > 3) goto bad;
      for (int f = 0; f < 3; f++)
        for (int q = 0; q < 4; q++)
          if (c->acc_fp[j].col[f][q] >= c->n_data) goto bad;
    }
  } else if ((c->code_cols || c->acc_cols) && (4 * c->n_acc != c->group_size[ORC_GROUP_ACCUM] || c->n_mix != 8 * c->n_acc)) {
    goto bad;
  }
  derive_regs_and_combos(c);
  return c;
bad:
  orc_circuit_free(c);
  return NULL;
}

/* ------------------------------------------------------------------ synthetic witness */
static inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
/* uniform-ish word in [0,p), used directly as a Montgomery word */
static inline fp_t synth_word(uint64_t seed, uint32_t stream, uint32_t row) {
  uint64_t h = splitmix64(splitmix64(seed) ^ (((uint64_t)stream << 32) | row));
  return (fp_t)(((h >> 32) * (uint64_t)ORC_P) >> 32);
}
#define CODE_SEED 0xC0DEull

void orc_witgen(const orc_circuit_t* c, uint32_t po2, uint64_t seed, uint32_t* code, uint32_t* data, uint32_t* global) {
  orc_witgen_public(c, po2, seed, NULL, code, data, global);
}

/* global_in != NULL: the caller's public inputs are planted at row 0 of the columns the globals are read from */
void orc_witgen_public(const orc_circuit_t* c, uint32_t po2, uint64_t seed, const uint32_t* global_in, uint32_t* code, uint32_t* data,
                       uint32_t* global) {
  orc_witgen_foreign_code(c, po2, seed, CODE_SEED, global_in, code, data, global);
}

/* A witness over CODE columns that are NOT the program's (fixed columns drawn from another seed, DATA derived from them so that
 * every constraint still holds): what a cheating prover would commit.  Tests use it to show that only the control-root
 * comparison of the verifier rejects such a seal. */
void orc_witgen_foreign_code(const orc_circuit_t* c, uint32_t po2, uint64_t seed, uint64_t code_seed, const uint32_t* global_in,
                             uint32_t* code, uint32_t* data, uint32_t* global) {
  size_t n = (size_t)1 << po2;
  for (uint32_t k = 0; k < c->n_code; k++) {
    fp_t* col = code + (size_t)k * n;
    uint32_t kind = c->code_cols[k].kind;
#pragma omp parallel for
    for (size_t r = 0; r < n; r++) {
      fp_t v = 0;
      if (kind == 0) v = r == 0 ? ORC_ONE : 0;
      else if (kind == 1) v = r == n - 1 ? ORC_ONE : 0;
      else if (kind == 2) v = fp_enc((uint32_t)r);
      else if (kind == 4) v = r < 65536 ? fp_enc((uint32_t)r) : 0;                                                        /* the 16-bit range table */
      else if (kind == 5) v = fp_enc(TAG_AND + (r < 65536 ? (uint32_t)r + 65536u * (((uint32_t)r & 255u) & ((uint32_t)r >> 8)) : 0u));  /* the byte-AND table */
      else if (kind == 6) v = r < (n / c->period) * c->period ? fp_enc(c->periodic[(size_t)c->code_cols[k].param * c->period + r % c->period]) : 0;  /* a periodic schedule */
      else v = synth_word(code_seed, (1u << 16) | k, (uint32_t)r);
      col[r] = v;
    }
  }
  /* a circuit with the in-circuit sponge: its columns hold the sponge over no words at all, and -- unless the caller names the public
   * inputs -- the inputs its digest is tied to are that digest (a caller who names them plants the rows of what it hashed instead) */
  uint32_t* sponge = NULL;
  if (c->has_sponge) {
    sponge = (uint32_t*)malloc(65 * n * sizeof(uint32_t));
    if (orc_sponge_trace(NULL, 0, po2, sponge) != 0) { free(sponge); sponge = NULL; }
  }
  for (uint32_t k = 0; k < c->n_data; k++) {
    fp_t* col = data + (size_t)k * n;
    const orc_data_col_t* d = &c->data_cols[k];
    if (d->kind == 0) {
      if (sponge && k >= c->sponge_data && k < c->sponge_data + 65) { memcpy(col, sponge + (size_t)(k - c->sponge_data) * n, n * sizeof(uint32_t)); continue; }
#pragma omp parallel for
      for (size_t r = 0; r < n; r++) col[r] = synth_word(seed, (2u << 16) | k, (uint32_t)r);
      for (uint32_t g = 0; g < c->n_global; g++)
        if (c->global_cols[g] == k) {
          if (global_in) col[0] = global_in[g];
          else if (sponge && g >= c->sponge_global && g < c->sponge_global + 8) col[0] = sponge[(size_t)(g - c->sponge_global) * n + 29];
        }
      continue;
    }
    const uint32_t refs[4] = {d->a, d->b, d->c, d->e};
    const fp_t* src[4]; uint32_t back[4];
    for (int i = 0; i < 4; i++) {
      src[i] = (REF_GROUP(refs[i]) == ORC_GROUP_CODE ? code : data) + (size_t)REF_COL(refs[i]) * n;
      back[i] = REF_BACK(refs[i]);
    }
#pragma omp parallel for
    for (size_t r = 0; r < n; r++) {
#define AT(i) src[i][(r + n - back[i]) & (n - 1)]
      fp_t prod = fp_mul(AT(0), AT(1));
      if (d->kind == 2) prod = fp_mul(prod, AT(2));
      col[r] = fp_add(prod, AT(3));
#undef AT
    }
  }
  free(sponge);
  const uint32_t* gcols = c->global_cols;
  for (uint32_t k = 0; k < c->n_global; k++) global[k] = data[(size_t)gcols[k] * n];
}

/* ---- the log-derivative argument (blob section 10; risc0-circuit-rv32im 4.0.4 `step_accum` stands here) */
static fp_t lf_eval(const orc_lf_t* lf, const uint32_t* code, const uint32_t* data, const uint32_t* global, size_t n, size_t r) {
  fp_t acc = 0;
  for (uint32_t t = 0; t < lf->n; t++) {
    fp_t v = fp_enc(lf->t[t].coef);
    if (lf->t[t].global) v = fp_mul(v, global[lf->t[t].global - 1]);
    if (lf->t[t].col) {
      uint32_t ref = lf->t[t].col - 1;
      const uint32_t* src = (ref >> 28) == ORC_GROUP_CODE ? code : data;
      v = fp_mul(v, src[(size_t)(ref & 0xfffffu) * n + r]);
    }
    acc = fp_add(acc, v);
  }
  return acc;
}
/* sum over the accumulator's four fractions of numerator / denominator on row r */
static fp4_t logup_term(const orc_logup_acc_t* a, const uint32_t* code, const uint32_t* data, const uint32_t* global, const uint32_t* mix, size_t n, size_t r) {
  fp4_t d[4];
  fp_t num[4];
  for (int f = 0; f < 4; f++) {
    const orc_logup_fraction_t* fr = &a->fr[f];
    num[f] = lf_eval(&fr->num, code, data, global, n, r);
    fp4_t den = fp4_zero();
    for (uint32_t k = 0; k < fr->n_parts; k++) {
      const orc_logup_part_t* q = &fr->parts[k];
      fp_t v = lf_eval(&q->lf, code, data, global, n, r);
      if (q->ch_kind == 0) den.e[0] = fp_add(den.e[0], v);
      else {
        fp4_t ch;
        memcpy(&ch, q->ch_kind == 1 ? mix + 4 * (size_t)q->ch_idx : global + q->ch_idx, 16);
        den = fp4_add(den, fp4_scale(ch, v));
      }
    }
    d[f] = den;
  }
  fp4_t d01 = fp4_mul(d[0], d[1]), d23 = fp4_mul(d[2], d[3]);
  fp4_t top = fp4_add(fp4_mul(fp4_add(fp4_scale(d[1], num[0]), fp4_scale(d[0], num[1])), d23), fp4_mul(fp4_add(fp4_scale(d[3], num[2]), fp4_scale(d[2], num[3])), d01));
  return fp4_mul(top, fp4_inv(fp4_mul(d01, d23)));
}

/* the multiplicity columns of DATA from the lookups the rows make (before DATA is committed); -1 if a value is not in its table */
int orc_logup_multiplicities(const orc_circuit_t* c, uint32_t po2, uint32_t* data, const uint32_t* global) {
  size_t n = (size_t)1 << po2;
  if (!c->n_tables) return 0;
  if (po2 < 16) return -1;
  uint32_t* hist = (uint32_t*)calloc((size_t)c->n_tables * 65536, 4);
  int bad = 0;
  for (uint32_t j = 0; j < c->n_chain; j++)
    for (int f = 0; f < 4; f++) {
      const orc_logup_fraction_t* fr = &c->logup[j].fr[f];
      if (!fr->table || fr->table > c->n_tables) continue;
      for (size_t r = 0; r < n; r++) {
        if (lf_eval(&fr->num, data, data, global, n, r) != ORC_ONE) continue;
        uint32_t v = fp_dec(fp_sub(0, lf_eval(&fr->parts[1].lf, data, data, global, n, r)));
        if (fr->table == 2) {
          v -= TAG_AND;
          if (v >> 24 || ((v & 255u) & ((v >> 8) & 255u)) != v >> 16) { bad = 1; continue; }
          v &= 0xffffu;
        } else if (v >> 16) { bad = 1; continue; }
        hist[(size_t)(fr->table - 1) * 65536 + v]++;
      }
    }
  for (uint32_t k = 0; k < c->n_tables; k++) {
    uint32_t* col = data + (size_t)c->table_col[k] * n;
    memset(col, 0, n * 4);
    for (uint32_t v = 0; v < 65536; v++) col[v] = fp_enc(hist[(size_t)k * 65536 + v]);
  }
  free(hist);
  return bad ? -1 : 0;
}

/* the totals of the accumulators that run alone (their challenges are public inputs): written into global_io where the circuit reads them */
void orc_logup_totals(const orc_circuit_t* c, uint32_t po2, const uint32_t* code, const uint32_t* data, uint32_t* global_io) {
  size_t n = (size_t)1 << po2;
  for (uint32_t j = c->n_chain; j < c->n_logup; j++) {
    fp4_t tot = fp4_zero();
#pragma omp parallel
    {
      fp4_t part = fp4_zero();
#pragma omp for nowait
      for (size_t r = 0; r < n; r++) part = fp4_add(part, logup_term(&c->logup[j], code, data, global_io, NULL, n, r));
#pragma omp critical
      tot = fp4_add(tot, part);
    }
    memcpy(global_io + c->logup[j].final_global, &tot, 16);
  }
}

static void logup_accum(const orc_circuit_t* c, uint32_t po2, const uint32_t* code, const uint32_t* data, const uint32_t* global, const uint32_t* mix, uint32_t* accum) {
  size_t n = (size_t)1 << po2;
  fp4_t* term = (fp4_t*)malloc(sizeof(fp4_t) * n * c->n_logup);
#pragma omp parallel for
  for (size_t r = 0; r < n; r++)
    for (uint32_t j = 0; j < c->n_logup; j++) term[r * c->n_logup + j] = logup_term(&c->logup[j], code, data, global, mix, n, r);
  /* the chain: one running sum through the row's links and on through the rows; the others run alone */
  fp4_t run = fp4_zero();
  for (size_t r = 0; r < n; r++)
    for (uint32_t j = 0; j < c->n_chain; j++) {
      run = fp4_add(run, term[r * c->n_logup + j]);
      for (int k = 0; k < 4; k++) accum[((size_t)4 * j + k) * n + r] = run.e[k];
    }
  for (uint32_t j = c->n_chain; j < c->n_logup; j++) {
    run = fp4_zero();
    for (size_t r = 0; r < n; r++) {
      run = fp4_add(run, term[r * c->n_logup + j]);
      for (int k = 0; k < 4; k++) accum[((size_t)4 * j + k) * n + r] = run.e[k];
    }
  }
  free(term);
}

void orc_accum_public(const orc_circuit_t* c, uint32_t po2, const uint32_t* code, const uint32_t* data, const uint32_t* global, const uint32_t* mix,
                      uint32_t* accum) {
  if (c->logup) logup_accum(c, po2, code, data, global, mix, accum);
  else orc_accum(c, po2, code, data, mix, accum);
}

void orc_accum(const orc_circuit_t* c, uint32_t po2, const uint32_t* code, const uint32_t* data, const uint32_t* mix,
               uint32_t* accum) {
  (void)code;
  size_t n = (size_t)1 << po2;
  fp4_t* tmp = (fp4_t*)malloc(sizeof(fp4_t) * n);
  for (uint32_t j = 0; j < c->n_acc_fp; j++) {
    /* running product over the rows of prod_f (alpha - addr_f - b1 lo_f - b2 hi_f - b3 t_f): the tuples one side of the
     * memory argument reads (or writes) in a row */
    const orc_acc_fp_t* a = &c->acc_fp[j];
    fp4_t m[4];
    memcpy(m, mix, 64);
#pragma omp parallel for
    for (size_t r = 0; r < n; r++) {
      fp4_t prod = fp4_one();
      for (uint32_t f = 0; f < a->n_f; f++) {
        const fp_t addr = data[(size_t)a->col[f][0] * n + r], lo = data[(size_t)a->col[f][1] * n + r];
        const fp_t hi = data[(size_t)a->col[f][2] * n + r], t = data[(size_t)a->col[f][3] * n + r];
        fp4_t term = fp4_sub(fp4_sub(fp4_sub(m[0], fp4_scale(m[1], lo)), fp4_scale(m[2], hi)), fp4_scale(m[3], t));
        term = fp4_sub(term, fp4_from_fp(addr));
        prod = fp4_mul(prod, term);
      }
      tmp[r] = prod;
    }
    orc_prefix_products((uint32_t*)tmp, (uint32_t)n);
    for (int k = 0; k < 4; k++) {
      fp_t* dst = accum + ((size_t)4 * j + k) * n;
      for (size_t r = 0; r < n; r++) dst[r] = tmp[r].e[k];
    }
  }
  for (uint32_t j = 0; j < c->n_acc; j++) {
    fp4_t m0, m1;
    memcpy(&m0, mix + 8 * j, 16); memcpy(&m1, mix + 8 * j + 4, 16);
    const fp_t* a = data + (size_t)c->acc_cols[j].a * n;
    const fp_t* b = data + (size_t)c->acc_cols[j].b * n;
#pragma omp parallel for
    for (size_t r = 0; r < n; r++) tmp[r] = fp4_add(fp4_add(m0, fp4_from_fp(a[r])), fp4_scale(m1, b[r]));
    orc_prefix_products((uint32_t*)tmp, (uint32_t)n);
    for (int k = 0; k < 4; k++) {
      fp_t* dst = accum + ((size_t)4 * j + k) * n;
      for (size_t r = 0; r < n; r++) dst[r] = tmp[r].e[k];
    }
  }
  free(tmp);
}

/* ------------------------------------------------------------------ constraint program */
typedef struct { fp4_t tot, mul; } mix_state_t;

/* The `mul` half of every MixState depends only on poly_mix: precompute once. */
static fp4_t* mix_muls(const orc_circuit_t* c, fp4_t poly_mix) {
  fp4_t* mul = (fp4_t*)malloc(sizeof(fp4_t) * (c->n_mix_vars ? c->n_mix_vars : 1));
  uint32_t m = 0;
  for (uint32_t i = 0; i < c->n_steps; i++) {
    const orc_step_t* s = &c->steps[i];
    if (s->op == OP_TRUE) mul[m++] = fp4_one();
    else if (s->op == OP_AND_EQZ) { mul[m] = fp4_mul(mul[s->a], poly_mix); m++; }
    else if (s->op == OP_AND_COND) { mul[m] = fp4_mul(mul[s->a], mul[s->c]); m++; }
  }
  return mul;
}

void orc_eval_check(const orc_circuit_t* c, uint32_t po2, const uint32_t* eval_accum, const uint32_t* eval_code,
                    const uint32_t* eval_data, const uint32_t* global, const uint32_t* mix, const uint32_t poly_mix[4],
                    uint32_t* check) {
  size_t n = (size_t)1 << po2, domain = n * ORC_INV_RATE;
  const uint32_t* groups[3] = {eval_accum, eval_code, eval_data};
  const uint32_t* globals[2] = {global, mix};
  fp4_t pm; memcpy(&pm, poly_mix, 16);
  fp4_t* mul = mix_muls(c, pm);
  /* (3 w^i)^N - 1 takes four values: 3^N * (w_4)^(i mod 4) - 1, w_4 = ROU_FWD[2] */
  fp_t inv_van[4];
  {
    fp_t three_n = fp_pow(fp_enc(3), n), w4 = orc_rou_fwd(2), cur = ORC_ONE;
    for (int k = 0; k < 4; k++) { inv_van[k] = fp_inv(fp_sub(fp_mul(three_n, cur), ORC_ONE)); cur = fp_mul(cur, w4); }
  }
#pragma omp parallel
  {
    fp_t* fv = (fp_t*)malloc(sizeof(fp_t) * (c->n_fp_vars ? c->n_fp_vars : 1));
    fp4_t* mt = (fp4_t*)malloc(sizeof(fp4_t) * (c->n_mix_vars ? c->n_mix_vars : 1));
#pragma omp for schedule(static)
    for (size_t i = 0; i < domain; i++) {
      uint32_t f = 0, m = 0;
      for (uint32_t k = 0; k < c->n_steps; k++) {
        const orc_step_t* s = &c->steps[k];
        switch (s->op) {
          case OP_CONST: fv[f++] = fp_enc(s->a); break;
          case OP_GET: {
            const orc_tap_t* t = &c->taps[s->a];
            fv[f++] = groups[t->group][(size_t)t->offset * domain + ((i + domain - ORC_INV_RATE * (size_t)t->back) & (domain - 1))];
            break;
          }
          case OP_GET_GLOBAL: fv[f++] = globals[s->a][s->b]; break;
          case OP_ADD: fv[f] = fp_add(fv[s->a], fv[s->b]); f++; break;
          case OP_SUB: fv[f] = fp_sub(fv[s->a], fv[s->b]); f++; break;
          case OP_MUL: fv[f] = fp_mul(fv[s->a], fv[s->b]); f++; break;
          case OP_TRUE: mt[m++] = fp4_zero(); break;
          case OP_AND_EQZ: mt[m] = fp4_add(mt[s->a], fp4_scale(mul[s->a], fv[s->b])); m++; break;
          case OP_AND_COND: mt[m] = fp4_add(mt[s->a], fp4_mul(fp4_scale(mt[s->c], fv[s->b]), mul[s->a])); m++; break;
          default: break;
        }
      }
      fp4_t r = fp4_scale(mt[c->ret], inv_van[i & 3]);
      for (int k = 0; k < 4; k++) check[(size_t)k * domain + i] = r.e[k];
    }
    free(fv); free(mt);
  }
  free(mul);
}

void orc_poly_ext(const orc_circuit_t* c, const uint32_t poly_mix[4], const uint32_t* u, const uint32_t* global,
                  const uint32_t* mix, uint32_t tot[4]) {
  const uint32_t* globals[2] = {global, mix};
  fp4_t pm; memcpy(&pm, poly_mix, 16);
  fp4_t* fv = (fp4_t*)malloc(sizeof(fp4_t) * (c->n_fp_vars ? c->n_fp_vars : 1));
  mix_state_t* mv = (mix_state_t*)malloc(sizeof(mix_state_t) * (c->n_mix_vars ? c->n_mix_vars : 1));
  uint32_t f = 0, m = 0;
  for (uint32_t k = 0; k < c->n_steps; k++) {
    const orc_step_t* s = &c->steps[k];
    switch (s->op) {
      case OP_CONST: fv[f++] = fp4_from_fp(fp_enc(s->a)); break;
      case OP_GET: memcpy(&fv[f++], u + 4 * s->a, 16); break;
      case OP_GET_GLOBAL: fv[f++] = fp4_from_fp(globals[s->a][s->b]); break;
      case OP_ADD: fv[f] = fp4_add(fv[s->a], fv[s->b]); f++; break;
      case OP_SUB: fv[f] = fp4_sub(fv[s->a], fv[s->b]); f++; break;
      case OP_MUL: fv[f] = fp4_mul(fv[s->a], fv[s->b]); f++; break;
      case OP_TRUE: mv[m].tot = fp4_zero(); mv[m].mul = fp4_one(); m++; break;
      case OP_AND_EQZ:
        mv[m].tot = fp4_add(mv[s->a].tot, fp4_mul(mv[s->a].mul, fv[s->b]));
        mv[m].mul = fp4_mul(mv[s->a].mul, pm);
        m++;
        break;
      case OP_AND_COND:
        mv[m].tot = fp4_add(mv[s->a].tot, fp4_mul(fp4_mul(fv[s->b], mv[s->c].tot), mv[s->a].mul));
        mv[m].mul = fp4_mul(mv[s->a].mul, mv[s->c].mul);
        m++;
        break;
      default: break;
    }
  }
  memcpy(tot, &mv[c->ret].tot, 16);
  free(fv); free(mv);
}
