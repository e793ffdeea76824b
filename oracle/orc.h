/* ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_field.h header).  Public C API of the CPU restatement.
 *
 * Every function restates one operation of the reference's prove_segment path, which lives in the
 * un-vendored crates risc0-zkp 3.0.4 / risc0-circuit-rv32im(-sys) 4.0.x / risc0-sys 1.5.0
 * (Cargo.lock:3195-3197, 3087-3089, 3121-3123, 3174-3176) reached from host/src/main.rs:420,423.
 * Citations in the .c files name the upstream module each function follows; they are recalled
 * ([R] in SURVEY.md), since no source for the path exists under /root/reference.
 *
 * PARITY STATUS: field, roots of unity and Poseidon2 parameters are pinned (cross-validated against the
 * published generation procedure + recalled risc0 words: tests/test_oracle_*.py).  The composed seal is
 * "parity unpinned" against risc0 itself: the reference commits only dev-mode Fake receipts
 * (data/test/test.xml-Receipt-test.json:1) and no Rust toolchain exists here.
 */
#ifndef ORC_H
#define ORC_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ORC_INV_RATE 4
#define ORC_QUERIES 50
#define ORC_FRI_FOLD 16
#define ORC_FRI_MIN_DEGREE 256
#define ORC_CHECK_SIZE 16
#define ORC_CELLS 24
#define ORC_RATE 16
#define ORC_DIGEST_WORDS 8
#define ORC_GROUP_ACCUM 0
#define ORC_GROUP_CODE 1
#define ORC_GROUP_DATA 2

void orc_set_threads(int n);
int orc_get_threads(void);

/* ---- field helpers exported for ctypes ---- */
uint32_t orc_fp_mul(uint32_t a, uint32_t b);
uint32_t orc_fp_enc(uint32_t canonical);
uint32_t orc_fp_dec(uint32_t mont);
uint32_t orc_fp_inv(uint32_t a);
uint32_t orc_fp_pow(uint32_t a, uint64_t n);
void orc_fp4_mul(const uint32_t a[4], const uint32_t b[4], uint32_t out[4]);
void orc_fp4_inv(const uint32_t a[4], uint32_t out[4]);
uint32_t orc_rou_fwd(unsigned po2); /* Montgomery form */
uint32_t orc_rou_rev(unsigned po2);

/* ---- NTT family (risc0-zkp core/ntt.rs, hal/cpu.rs) ---- */
void orc_batch_interpolate_ntt(uint32_t* io, uint32_t count, uint32_t po2);
void orc_batch_expand_into_evaluate_ntt(uint32_t* out, const uint32_t* in, uint32_t count, uint32_t in_po2,
                                        uint32_t expand_bits);
void orc_batch_bit_reverse(uint32_t* io, uint32_t count, uint32_t po2);
void orc_zk_shift(uint32_t* io, uint32_t count, uint32_t po2);

/* ---- Poseidon2 (risc0-zkp core/hash/poseidon2) ---- */
void orc_poseidon2_consts(uint32_t* rc_canonical /*24*29*/, uint32_t* diag_m1_canonical /*24*/);
void orc_poseidon2_mix(uint32_t cells[ORC_CELLS]);
void orc_hash_elem_slice(const uint32_t* elems, size_t n, uint32_t digest[8]);
int orc_sponge_trace(const uint32_t* words, size_t n_words, uint32_t po2, uint32_t* cols /* [65][2^po2] */);
void orc_hash_pair(const uint32_t a[8], const uint32_t b[8], uint32_t out[8]);
void orc_hash_rows(uint32_t* digests, const uint32_t* matrix, size_t rows, size_t cols);
void orc_hash_fold(uint32_t* nodes, size_t output_size); /* nodes[o..2o) = H(nodes[2i],nodes[2i+1]) */

/* ---- Merkle (risc0-zkp prove/merkle.rs) ---- */
typedef struct {
  size_t row_size, col_size, queries, layers, top_layer, top_size;
} orc_merkle_params_t;
void orc_merkle_params(orc_merkle_params_t* p, size_t row_size, size_t col_size, size_t queries);
void orc_merkle_build(uint32_t* nodes /* 2*row_size*8 */, const uint32_t* matrix, size_t row_size, size_t col_size);

/* ---- streaming ops (risc0-zkp hal/cpu.rs) ---- */
void orc_batch_evaluate_any(const uint32_t* coeffs, uint32_t po2, const uint32_t* which, const uint32_t* xs /*ext*/,
                            uint32_t n_eval, uint32_t* out /*ext*/);
void orc_mix_poly_coeffs(uint32_t* combos /*ext AoS [n_combo][N]*/, const uint32_t mix_start[4], const uint32_t mix[4],
                         const uint32_t* input, const uint32_t* combo_of, uint32_t input_count, uint32_t po2);
void orc_eltwise_sum_extelem(uint32_t* out /*[4][N]*/, const uint32_t* in /*ext AoS [count][N]*/, uint32_t count,
                             uint32_t n);
void orc_fri_fold(uint32_t* out /*[4][n/16]*/, const uint32_t* in /*[4][n]*/, const uint32_t mix[4], uint32_t n_out);
void orc_prefix_products(uint32_t* io /*ext AoS*/, uint32_t n);
void orc_poly_divide(uint32_t* poly /*ext AoS n*/, uint32_t n, const uint32_t z[4], uint32_t rem[4]);
void orc_poly_interpolate(uint32_t* out /*ext n*/, const uint32_t* xs, const uint32_t* ys, uint32_t n);

/* ---- circuit blob, witgen, accum, eval_check (risc0-circuit-rv32im + -sys; data-driven here) ---- */
typedef struct orc_circuit orc_circuit_t;
orc_circuit_t* orc_circuit_parse(const uint32_t* blob, size_t n_words);
void orc_circuit_free(orc_circuit_t* c);
uint32_t orc_circuit_group_size(const orc_circuit_t* c, uint32_t group);
uint32_t orc_circuit_n_taps(const orc_circuit_t* c);
uint32_t orc_circuit_n_global(const orc_circuit_t* c);
uint32_t orc_circuit_n_mix(const orc_circuit_t* c);
uint32_t orc_circuit_n_combos(const orc_circuit_t* c);
void orc_witgen(const orc_circuit_t* c, uint32_t po2, uint64_t seed, uint32_t* code, uint32_t* data, uint32_t* global);
void orc_witgen_public(const orc_circuit_t* c, uint32_t po2, uint64_t seed, const uint32_t* global_in, uint32_t* code, uint32_t* data,
                       uint32_t* global);
void orc_witgen_foreign_code(const orc_circuit_t* c, uint32_t po2, uint64_t seed, uint64_t code_seed, const uint32_t* global_in,
                             uint32_t* code, uint32_t* data, uint32_t* global);
void orc_accum(const orc_circuit_t* c, uint32_t po2, const uint32_t* code, const uint32_t* data, const uint32_t* mix,
               uint32_t* accum);
/* the log-derivative argument of a circuit (blob section 10): multiplicity columns of DATA (returns -1 if a looked-up value is in no
 * table), totals of the accumulators whose challenges are public inputs, and the accumulation with public inputs at hand */
int orc_logup_multiplicities(const orc_circuit_t* c, uint32_t po2, uint32_t* data, const uint32_t* global);
void orc_logup_totals(const orc_circuit_t* c, uint32_t po2, const uint32_t* code, const uint32_t* data, uint32_t* global_io);
void orc_accum_public(const orc_circuit_t* c, uint32_t po2, const uint32_t* code, const uint32_t* data, const uint32_t* global,
                      const uint32_t* mix, uint32_t* accum);
uint32_t orc_circuit_n_late(const orc_circuit_t* c);
void orc_eval_check(const orc_circuit_t* c, uint32_t po2, const uint32_t* eval_accum, const uint32_t* eval_code,
                    const uint32_t* eval_data, const uint32_t* global, const uint32_t* mix, const uint32_t poly_mix[4],
                    uint32_t* check /*[4][4N]*/);
/* Evaluate the constraint program on extension-field tap values (the verifier's view). */
void orc_poly_ext(const orc_circuit_t* c, const uint32_t poly_mix[4], const uint32_t* u /*ext per tap*/,
                  const uint32_t* global, const uint32_t* mix, uint32_t tot[4]);

/* ---- sequencer + verifier (risc0-zkp prove/prover.rs, prove/fri.rs, verify/mod.rs, verify/fri.rs) ---- */
/* Returns seal length in words (0 on failure). If seal==NULL only the length is computed (still proves). */
size_t orc_prove_segment(const orc_circuit_t* c, const uint32_t* blob, size_t blob_words, uint32_t po2,
                         const uint32_t* code, const uint32_t* data, const uint32_t* global, uint32_t* seal,
                         size_t seal_cap);
/* 0 = accepted; otherwise a positive error code naming the first failed check. */
int orc_verify_segment(const orc_circuit_t* c, const uint32_t* blob, size_t blob_words, const uint32_t* seal,
                       size_t seal_words);
/* the same, with the CODE commitment compared against the program's control root (8 words; NULL = not compared) */
int orc_verify_segment_bound(const orc_circuit_t* c, const uint32_t* blob, size_t blob_words, const uint32_t* seal,
                             size_t seal_words, const uint32_t* expected_code_root);
/* control root of a CODE group: iNTT + zk shift, expand x4, Merkle root -- as commit_group does it */
void orc_code_root(const uint32_t* code, uint32_t count, uint32_t po2, uint32_t root[8]);
const char* orc_verify_strerror(int code);

/* ---- Fiat-Shamir transcript (risc0-zkp core/hash/poseidon2/rng.rs, prove/write_iop.rs) ---- */
typedef struct {
  uint32_t cells[ORC_CELLS];
  uint32_t pool_used;
} orc_rng_t;
void orc_rng_init(orc_rng_t* r);
void orc_rng_mix(orc_rng_t* r, const uint32_t digest[8]);
uint32_t orc_rng_elem(orc_rng_t* r);
uint32_t orc_rng_bits(orc_rng_t* r, uint32_t bits);

#ifdef __cplusplus
}
#endif
#endif
