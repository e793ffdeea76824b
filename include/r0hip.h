/* r0hip -- C ABI of the MI355X-native RISC Zero segment prover (hot path only: prove_segment).
 *
 * This is the drop-in boundary for the path the reference reaches through
 *     host/src/main.rs:420   let prover = default_prover();
 *     host/src/main.rs:423   prover.prove(env, HYPERFRIDGE_ELF)
 * i.e. (inside the un-vendored risc0 crates pinned in Cargo.lock:3195-3197, 3121-3123, 3174-3176) the
 * `risc0_zkp::hal::Hal` + `CircuitHal` operations that a native backend implements and that risc0's own
 * CUDA/Metal backends bind as `extern "C"` launchers from risc0-sys / risc0-circuit-rv32im-sys.
 * A Rust `HipHal` would bind exactly these entry points (INTEGRATION.md shows the stub).
 *
 * Conventions (mirroring risc0-sys's launcher convention):
 *   - every function returns `const char*`: NULL on success, otherwise a heap-allocated message the caller
 *     releases with r0h_free_error().  Nothing aborts or throws across the ABI.
 *   - field elements are uint32_t BabyBear words in Montgomery form (R = 2^32), always < p = 2013265921;
 *     extension elements are 4 consecutive words (x^4 = 11); digests are 8 words.
 *   - matrices are column-major: column c of a [count][size] buffer starts at word c*size.
 *   - a context is bound to one device and one stream and is not re-entrant; different contexts are independent.
 *     Operations are stream-ordered; only r0h_buf_d2h, r0h_sync and the functions that return host data block.
 *   - the caller owns host memory; the library owns device memory behind r0h_buf handles.
 */
#ifndef R0HIP_H
#define R0HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define R0H_P 2013265921u
#define R0H_INV_RATE 4
#define R0H_QUERIES 50
#define R0H_FRI_FOLD 16
#define R0H_FRI_MIN_DEGREE 256
#define R0H_CHECK_SIZE 16
#define R0H_DIGEST_WORDS 8
#define R0H_GROUP_ACCUM 0
#define R0H_GROUP_CODE 1
#define R0H_GROUP_DATA 2
#define R0H_MAX_PO2 24 /* trace rows of a segment: risc0's default segment size is 2^20, its largest 2^24.  The evaluation domain is
                        * 4x that and the NTT entry points take up to 2^26 points: two passes up to 2^23, three above
                        * (a 256-column segment of 2^22 rows keeps about 32 GiB resident, one of 2^24 rows about 128 GiB) */

typedef struct r0h_ctx r0h_ctx;
typedef struct r0h_buf r0h_buf;
typedef struct r0h_circuit r0h_circuit;

void r0h_free_error(const char* msg);
const char* r0h_version(void);

/* ---- context and buffers: Hal::alloc_*, copy_from_*, Buffer::{slice, view} ---- */
const char* r0h_ctx_create(int device, r0h_ctx** out);
const char* r0h_ctx_destroy(r0h_ctx* ctx);
const char* r0h_sync(r0h_ctx* ctx);
const char* r0h_buf_alloc(r0h_ctx* ctx, size_t bytes, r0h_buf** out);
const char* r0h_buf_wrap(r0h_ctx* ctx, void* device_ptr, size_t bytes, r0h_buf** out); /* memory owned elsewhere */
const char* r0h_buf_slice(r0h_buf* parent, size_t offset_bytes, size_t bytes, r0h_buf** out);
const char* r0h_buf_free(r0h_buf* buf);
const char* r0h_buf_h2d(r0h_ctx* ctx, r0h_buf* dst, size_t offset_bytes, const void* src, size_t bytes);
const char* r0h_buf_d2h(r0h_ctx* ctx, const r0h_buf* src, size_t offset_bytes, void* dst, size_t bytes);
const char* r0h_buf_zero(r0h_ctx* ctx, r0h_buf* buf);
void* r0h_buf_device_ptr(const r0h_buf* buf);
size_t r0h_buf_bytes(const r0h_buf* buf);

/* ---- Hal: NTT family (risc0-zkp hal `batch_interpolate_ntt`, `batch_expand_into_evaluate_ntt`,
 *      `batch_bit_reverse`, `zk_shift`) ---- */
/* count columns of 2^po2 natural-order evaluations -> bit-reversed coefficients, in place */
const char* r0h_batch_interpolate_ntt(r0h_ctx* ctx, r0h_buf* io, uint32_t count, uint32_t po2);
/* count columns of 2^in_po2 bit-reversed coefficients -> natural-order evaluations on the 2^(in_po2+expand_bits) domain */
const char* r0h_batch_expand_into_evaluate_ntt(r0h_ctx* ctx, r0h_buf* out, const r0h_buf* in, uint32_t count,
                                               uint32_t in_po2, uint32_t expand_bits);
const char* r0h_batch_bit_reverse(r0h_ctx* ctx, r0h_buf* io, uint32_t count, uint32_t po2);
/* coefficient at bit-reversed position i is multiplied by 3^brev(i): f(x) -> f(3x) */
const char* r0h_zk_shift(r0h_ctx* ctx, r0h_buf* io, uint32_t count, uint32_t po2);
/* `batch_interpolate_ntt` followed by `zk_shift` as the prover issues them (prove/prover.rs commit_group), in one call: the shift
 * rides on the last pass of the inverse transform instead of sweeping the coefficients again.  Same words as the two calls. */
const char* r0h_batch_interpolate_ntt_zk_shift(r0h_ctx* ctx, r0h_buf* io, uint32_t count, uint32_t po2);

/* ---- Hal: Poseidon2 Merkle commitment (`hash_rows`, `hash_fold`; prove/merkle.rs) ---- */
/* rc: 24*29 canonical round constants, diag_m1: 24 canonical (mu_i - 1); the default table is compiled in */
const char* r0h_poseidon2_set_consts(r0h_ctx* ctx, const uint32_t* rc, const uint32_t* diag_m1);
/* digests[r] = sponge(matrix[0][r], matrix[1][r], ..., matrix[cols-1][r]) for r < rows */
const char* r0h_hash_rows(r0h_ctx* ctx, r0h_buf* digests, const r0h_buf* matrix, uint32_t rows, uint32_t cols);
/* nodes[i] = H(nodes[2i] || nodes[2i+1]) for output_size <= i < 2*output_size (digest units) */
const char* r0h_hash_fold(r0h_ctx* ctx, r0h_buf* nodes, uint32_t output_size);
/* nodes has 2*rows digests: leaves at [rows, 2rows), root at index 1 */
const char* r0h_merkle_build(r0h_ctx* ctx, r0h_buf* nodes, const r0h_buf* matrix, uint32_t rows, uint32_t cols);

/* ---- Hal: streaming ops (`batch_evaluate_any`, `mix_poly_coeffs`, `eltwise_*`, `gather_sample`, `scatter`,
 *      `fri_fold`, `prefix_products`; core/poly.rs `poly_divide`) ---- */
/* out[k] = poly which[k] (2^po2 natural-order coefficients) evaluated at the extension point xs[4k..4k+4);
 * which / xs are small host arrays (upstream uploads them from the host as well) */
const char* r0h_batch_evaluate_any(r0h_ctx* ctx, const r0h_buf* coeffs, uint32_t po2, const uint32_t* which_host,
                                   const uint32_t* xs_host, uint32_t n_eval, r0h_buf* out);
const char* r0h_mix_poly_coeffs(r0h_ctx* ctx, r0h_buf* combos, const uint32_t mix_start[4], const uint32_t mix[4],
                                const r0h_buf* input, const uint32_t* combo_of_host, uint32_t input_count,
                                uint32_t po2);
const char* r0h_eltwise_add_elem(r0h_ctx* ctx, r0h_buf* out, const r0h_buf* a, const r0h_buf* b, uint32_t n);
const char* r0h_eltwise_copy_elem(r0h_ctx* ctx, r0h_buf* out, const r0h_buf* in, uint32_t n);
const char* r0h_eltwise_sum_extelem(r0h_ctx* ctx, r0h_buf* out, const r0h_buf* in, uint32_t count, uint32_t n);
/* cells still holding Elem::INVALID (0xffffffff) become zero (risc0 `eltwise_zeroize_elem`, run on the witness before commit) */
const char* r0h_eltwise_zeroize_elem(r0h_ctx* ctx, r0h_buf* io, uint32_t n);
const char* r0h_gather_sample(r0h_ctx* ctx, r0h_buf* dst, const r0h_buf* src, uint32_t idx, uint32_t size,
                              uint32_t stride);
const char* r0h_scatter(r0h_ctx* ctx, r0h_buf* into, const r0h_buf* index, const r0h_buf* offsets,
                        const r0h_buf* values, uint32_t n_index);
const char* r0h_fri_fold(r0h_ctx* ctx, r0h_buf* out, const r0h_buf* in, const uint32_t mix[4], uint32_t n_out);
/* The same operations with the operand placement of the risc0-zkp `Hal` trait (as recalled): `which` (u32), `xs` (extension
 * elements) and `combos` (u32) are device Buffers there, `scatter` takes host slices, `hash_fold` names both level sizes.  A Rust
 * `HipHal` calls these directly; the grouping the kernels rely on is made on the host, so the Buffer forms read the small index
 * buffers back themselves (one copy of a few KB). */
const char* r0h_batch_evaluate_any_buf(r0h_ctx* ctx, const r0h_buf* coeffs, uint32_t po2, const r0h_buf* which, const r0h_buf* xs,
                                       uint32_t n_eval, r0h_buf* out);
const char* r0h_mix_poly_coeffs_buf(r0h_ctx* ctx, r0h_buf* combos, const uint32_t mix_start[4], const uint32_t mix[4],
                                    const r0h_buf* input, const r0h_buf* combo_of, uint32_t input_count, uint32_t po2);
/* values[k] goes to into[offsets[k]] for index[0] <= k < index[n_index - 1] (offsets / values hold n_values host words) */
const char* r0h_scatter_slices(r0h_ctx* ctx, r0h_buf* into, const uint32_t* index, uint32_t n_index, const uint32_t* offsets,
                               const uint32_t* values, uint32_t n_values);
const char* r0h_hash_fold_io(r0h_ctx* ctx, r0h_buf* io, uint32_t input_size, uint32_t output_size);
const char* r0h_prefix_products(r0h_ctx* ctx, r0h_buf* io, uint32_t n);
const char* r0h_prefix_sums(r0h_ctx* ctx, r0h_buf* io, uint32_t n); /* io[i] = sum_{j <= i} io[j] over extension elements (AoS), n a power of two */
/* in-place synthetic division of an extension-coefficient polynomial (AoS, natural order) by (x - z) */
const char* r0h_poly_divide(r0h_ctx* ctx, r0h_buf* poly, uint32_t n, const uint32_t z[4], uint32_t remainder[4]);

/* ---- CircuitHal: circuit blob (format: r0hip_circuit.h), witness generation, accumulation, eval_check ---- */
/* Pure host: emit the HIP source of the circuit's eval_check kernels (caller frees with r0h_free_error). */
const char* r0h_circuit_emit_hip(const uint32_t* blob, size_t n_words, char** source_out);
/* code_object_path: a gfx950 code object built from r0h_circuit_emit_hip's output, or NULL to compile in-process. */
const char* r0h_circuit_load(r0h_ctx* ctx, const uint32_t* blob, size_t n_words, const char* code_object_path,
                             r0h_circuit** out);
const char* r0h_circuit_free(r0h_circuit* c);
uint32_t r0h_circuit_group_size(const r0h_circuit* c, uint32_t group);
uint32_t r0h_circuit_n_global(const r0h_circuit* c);
uint32_t r0h_circuit_n_mix(const r0h_circuit* c);
uint32_t r0h_circuit_n_taps(const r0h_circuit* c);
const char* r0h_witgen(r0h_ctx* ctx, const r0h_circuit* c, uint32_t po2, uint64_t seed, r0h_buf* code, r0h_buf* data,
                       uint32_t* global_out_host);
/* the same with the public inputs chosen by the caller: global_in[k] (a canonical Montgomery word) is planted at row 0 of the
 * column global k is read from, before the dependent columns are derived -- how a recursion-shaped segment is bound to the
 * digests of the seals it stands for (hyperfridge-r0_amd/recursion.py) */
const char* r0h_witgen_public(r0h_ctx* ctx, const r0h_circuit* c, uint32_t po2, uint64_t seed, const uint32_t* global_in_host,
                              r0h_buf* code, r0h_buf* data);
const char* r0h_accum(r0h_ctx* ctx, const r0h_circuit* c, uint32_t po2, const r0h_buf* code, const r0h_buf* data,
                      const uint32_t* mix_host, r0h_buf* accum);
const char* r0h_eval_check(r0h_ctx* ctx, const r0h_circuit* c, uint32_t po2, const r0h_buf* eval_accum,
                           const r0h_buf* eval_code, const r0h_buf* eval_data, const uint32_t* global_host,
                           const uint32_t* mix_host, const uint32_t poly_mix[4], r0h_buf* check);

/* ---- the sequencer: risc0-circuit-rv32im `SegmentProver::prove` + risc0-zkp `Prover::{commit_group, finalize}` ---- */
/* code/data: witness columns resident in device memory ([group_size][2^po2]); global: host words.
 * seal_out receives *seal_words_out words (error if it exceeds seal_capacity_words). */
const char* r0h_prove_segment(r0h_ctx* ctx, const r0h_circuit* c, uint32_t po2, const r0h_buf* code,
                              const r0h_buf* data, const uint32_t* global_host, uint32_t* seal_out,
                              size_t seal_capacity_words, size_t* seal_words_out);
/* The same sequencer split around the accumulation step, as risc0-zkp's `Prover` is used by the segment driver
 * (commit_group(CODE), commit_group(DATA), draw the mix -> caller accumulates -> commit_group(ACCUM), finalize):
 * r0h_proof_begin commits CODE and DATA and returns the n_mix accumulation-mix words; the caller fills the ACCUM witness
 * ([group_size(ACCUM)][2^po2], e.g. with its own step_accum or r0h_accum) and hands it to r0h_proof_finish, which consumes
 * the proof object.  r0h_prove_segment == begin + r0h_accum + finish. */
typedef struct r0h_proof r0h_proof;
const char* r0h_proof_begin(r0h_ctx* ctx, const r0h_circuit* c, uint32_t po2, const r0h_buf* code, const r0h_buf* data,
                            const uint32_t* global_host, uint32_t* mix_out, r0h_proof** out);
const char* r0h_proof_finish(r0h_proof* proof, const r0h_buf* accum, uint32_t* seal_out, size_t seal_capacity_words,
                             size_t* seal_words_out);
const char* r0h_proof_abort(r0h_proof* proof);
/* A proof waiting between its phases holds its DATA group three ways: coefficients, evaluations on the 4N coset (4/5 of the bytes:
 * 2 GiB for 128 columns of 2^20 rows) and Merkle nodes.  r0h_proof_shrink gives the evaluations back to the context's pool;
 * r0h_proof_finish computes them again from the coefficients (one expanding NTT, the same words: the seal does not change).  A
 * session does this by itself for the segments beyond its resident limit (r0h_ctx_set_session_resident_limit). */
const char* r0h_proof_shrink(r0h_proof* proof, size_t* bytes_freed_out);
size_t r0h_proof_resident_bytes(const r0h_proof* proof);
/* Late public inputs (blob section LATE: inputs that depend on commitments made outside this proof -- the trace circuit's session
 * challenge, derived from the DATA roots of ALL segments, and the segment's sum under it).  For such a circuit r0h_proof_begin stops
 * after the DATA commitment without drawing the mix: r0h_proof_data_root gives the root, r0h_proof_late takes the last n_late
 * public inputs (the transcript absorbs them; the seal's opening block receives them) and draws the accumulation mix into mix_out.
 * r0h_prove_segment[_committed] takes all public inputs at once and does the same internally.  r0h_proof_globals: all of them, as
 * the proof holds them. */
const char* r0h_proof_data_root(const r0h_proof* proof, uint32_t root_out[8]);
const char* r0h_proof_late(r0h_proof* proof, const uint32_t* late_globals, uint32_t* mix_out);
const char* r0h_proof_globals(const r0h_proof* proof, uint32_t* globals_out);
/* Control root of a program at trace size 2^po2: Merkle root of the committed CODE group (count columns of 2^po2 words),
 * computed exactly as the sequencer commits it.  What a verifier passes to r0h_verify_seal_bound. */
const char* r0h_code_root(r0h_ctx* ctx, const r0h_buf* code, uint32_t count, uint32_t po2, uint32_t root_out[8]);
/* The CODE group depends on (circuit, po2) only -- its Merkle root is the control root risc0 tabulates per po2 -- so it is committed
 * once and kept: bit-reversed zk-shifted coefficients, evaluations on the 4N coset, Merkle nodes, top layer and root.  Every proof of
 * that size on any context of the same device reads it (nothing writes it), and its seal is word for word the seal of
 * r0h_prove_segment on the same CODE columns.  r0h_code_commit_new blocks until the commitment is complete; free it only after
 * the proofs that use it have returned.  r0h_prove_segment == r0h_code_commit_new + r0h_prove_segment_committed. */
typedef struct r0h_code_commit r0h_code_commit;
const char* r0h_code_commit_new(r0h_ctx* ctx, const r0h_buf* code, uint32_t count, uint32_t po2, r0h_code_commit** out);
const char* r0h_code_commit_free(r0h_code_commit* cc);
const char* r0h_code_commit_root(const r0h_code_commit* cc, uint32_t root_out[8]); /* == r0h_code_root of the same columns */
const char* r0h_code_commit_columns(const r0h_code_commit* cc, const r0h_buf** columns_out); /* the committed columns themselves (device; owned by cc) */
const char* r0h_prove_segment_committed(r0h_ctx* ctx, const r0h_circuit* c, uint32_t po2, const r0h_code_commit* code,
                                        const r0h_buf* data, const uint32_t* global_host, uint32_t* seal_out,
                                        size_t seal_capacity_words, size_t* seal_words_out);
const char* r0h_proof_begin_committed(r0h_ctx* ctx, const r0h_circuit* c, uint32_t po2, const r0h_code_commit* code,
                                      const r0h_buf* data, const uint32_t* global_host, uint32_t* mix_out, r0h_proof** out);
/* Per-phase device time of the last r0h_prove_segment on this context (ms), for bench.py; names are static strings. */
const char* r0h_last_profile(r0h_ctx* ctx, const char*** names_out, const float** ms_out, uint32_t* n_out);

/* ---- verifier: risc0-zkp verify/mod.rs as reached from `receipt.verify(image_id)` (host/src/main.rs:622-624,
 * verifier/src/main.rs:124-126).  Pure host code: no context, no GPU.  Returns NULL when the check itself ran (the outcome is
 * in *verdict_out: R0H_VERIFY_OK or the first reason for rejection) and an error string only for unusable arguments
 * (malformed circuit blob, non-canonical Poseidon2 tables).  p2_round_constants [29][24] / p2_diag_m1 [24] are canonical words,
 * or both NULL for the compiled-in risc0 table.  *po2_out (optional) receives the trace size the seal claims. ---- */
#define R0H_VERIFY_OK 0
#define R0H_VERIFY_TRUNCATED 1
#define R0H_VERIFY_BAD_PO2 2
#define R0H_VERIFY_MERKLE_GROUP 3
#define R0H_VERIFY_CHECK_MISMATCH 4
#define R0H_VERIFY_FRI_MERKLE 5
#define R0H_VERIFY_FRI_GOAL 6
#define R0H_VERIFY_FRI_FINAL 7
#define R0H_VERIFY_TRAILING 8
#define R0H_VERIFY_BAD_ELEM 9
#define R0H_VERIFY_CODE_ROOT 10
const char* r0h_verify_seal(const uint32_t* blob, size_t blob_words, const uint32_t* p2_round_constants,
                            const uint32_t* p2_diag_m1, const uint32_t* seal, size_t seal_words, int* verdict_out,
                            uint32_t* po2_out);
/* The same check, bound to a program: risc0-zkp verify/mod.rs calls `check_code(po2, root)` on the CODE commitment, which the
 * rv32im verifier answers from its table of control roots (one per po2).  expected_code_root (8 canonical words, e.g. from
 * r0h_code_root) is compared with the root the seal commits to; a mismatch is R0H_VERIFY_CODE_ROOT.  NULL skips the comparison
 * (then nothing ties the seal to a program: r0h_verify_seal is that form and is meant for tests of the proof system alone).
 * code_root_out (optional) receives the root found in the seal. */
const char* r0h_verify_seal_bound(const uint32_t* blob, size_t blob_words, const uint32_t* p2_round_constants,
                                  const uint32_t* p2_diag_m1, const uint32_t* seal, size_t seal_words,
                                  const uint32_t* expected_code_root, int* verdict_out, uint32_t* po2_out,
                                  uint32_t* code_root_out);
/* the same, also giving the DATA group's Merkle root as the seal's transcript recomputes it (the session challenge of the trace
 * circuit is derived from these roots: csrc/claim.cpp) */
const char* r0h_verify_seal_roots(const uint32_t* blob, size_t blob_words, const uint32_t* seal, size_t seal_words,
                                  const uint32_t* expected_code_root, int* verdict_out, uint32_t* po2_out, uint32_t* data_root_out);
/* The control root of a circuit's own CODE columns at trace size 2^po2, computed on the HOST from the blob alone (circuits with a
 * column program: the CODE columns r0h_witgen generates) -- the same 8 words as r0h_code_root / r0h_code_commit_root on the device.
 * A verifier that has the circuit need not be told its control roots (risc0's verifier has them compiled in).  Seconds at po2 = 20. */
const char* r0h_control_root_host(const uint32_t* blob, size_t blob_words, const uint32_t* p2_round_constants, const uint32_t* p2_diag_m1,
                                  uint32_t po2, uint32_t root_out[8]);
const char* r0h_verify_reason(int verdict); /* static string, do not free */
/* Poseidon2 sponge (compiled-in table) over a seal's words (all canonical field elements, else an error): the 8-word name a
 * recursion step commits to */
const char* r0h_seal_digest(const uint32_t* seal, size_t seal_words, uint32_t digest_out[8]);
/* The same sponge laid out as the rows of the recursion circuit's in-circuit hash (r0hip_circuit.h SPONGE; tools/sponge_component.py):
 * 65 columns (st[24], aux[24], in[16], act) of 2^po2 rows, 30 rows per permutation, zero behind the last one.  Host code; r0h_lift /
 * r0h_join plant exactly this into a node's witness, so that the digest among the node's public inputs is computed inside its proof.
 * Refused: words that are not canonical field elements, more words than the trace has rows for (30 rows per 16 words). */
const char* r0h_sponge_trace(const uint32_t* words, size_t n_words, uint32_t po2, uint32_t* cols_out /* [65][2^po2] */);

/* ---- data formats either side of the path (SURVEY.md 8(a) a0', a0'', a18): pure host code ----
 * serde word stream of a String: [u32 LE length][utf8][zero padding to 4] -- what `ExecutorEnv::builder().write(&s)` feeds the
 * guest (host/src/main.rs:389-417) and what `env::commit(&String)` leaves in `receipt.journal.bytes` (host/src/main.rs:258-267).
 * r0h_serde_encode_str with out == NULL only reports the size. */
const char* r0h_serde_encode_str(const uint8_t* utf8, size_t len, uint8_t* out, size_t capacity, size_t* out_len);
const char* r0h_serde_decode_str(const uint8_t* bytes, size_t n, size_t* str_off, size_t* str_len, size_t* consumed /* may be NULL */);
/* The input stream `ExecutorEnv::builder().write(..)` builds (host/src/main.rs:389-417: 12 Strings and one Vec<u8>), as the u32
 * words the guest's `env::read()` calls consume (methods/guest/src/main.rs:159-171).  A Vec<u8> goes one word per byte. */
typedef struct r0h_env r0h_env;
const char* r0h_env_new(r0h_env** out);
const char* r0h_env_write_str(r0h_env* e, const uint8_t* utf8, size_t len);
const char* r0h_env_write_u8_seq(r0h_env* e, const uint8_t* bytes, size_t len);
const char* r0h_env_words(const r0h_env* e, const uint32_t** words, size_t* n_words);
const char* r0h_env_free(r0h_env* e);
/* hyperfridge's reading of the commitment in a journal: first '{' .. last '}' (host/src/main.rs:258-267, verifier/src/main.rs:176-185) */
const char* r0h_journal_commitment_span(const uint8_t* bytes, size_t n, size_t* off, size_t* len);
/* Receipt JSON as `serde_json::to_string(&receipt)` writes it (host/src/main.rs:251-252) and `serde_json::from_slice` reads it
 * (verifier/src/main.rs:118-119).  Pinned by the reference's fixtures: {"inner":"Fake","journal":{"bytes":[u8..]}}, re-serialised
 * byte for byte.  The composite form follows risc0-zkvm 3.x as recalled (unpinned):
 *   {"inner":{"Composite":{"segments":[{"seal":[u32..],"index":n,"hashfn":"poseidon2","verifier_parameters":"<hex>","claim":
 *     {"pre":{"Value":{"pc":n,"merkle_root":"<hex>"}},"post":{..},"exit_code":{"Halted":0}|{"Paused":0}|"SystemSplit"|"SessionLimit",
 *      "input":{"Pruned":"<hex>"},"output":{"Pruned":"<hex>"}|{"Value":null}}},..],"assumption_receipts":[],"verifier_parameters":"<hex>"}},
 *    "journal":{"bytes":[..]},"metadata":{"verifier_parameters":"<hex>"}}
 * (serde_json is human-readable, so a risc0 `Digest` is a hex string).  Segments without a claim (this library's round-1 files) are
 * still read.  r0h_receipt_to_json's output is serde_json's compact form (caller frees it with r0h_free_error). */
#define R0H_RECEIPT_FAKE 0
#define R0H_RECEIPT_COMPOSITE 1
typedef struct r0h_receipt r0h_receipt;
/* risc0-binfmt `SystemState` and risc0-zkvm `ReceiptClaim`, flattened (field names, tags and digest layout are recalled from the
 * public risc0 sources: unpinned, see csrc/claim.cpp).  exit_system / exit_user: Halted(u) = (0, u), Paused(u) = (1, u),
 * SystemSplit = (2, 0), SessionLimit = (2, 2).  input_digest / output_digest: Digest::ZERO stands for None. */
typedef struct { uint32_t pc; uint8_t merkle_root[32]; } r0h_system_state;
typedef struct {
  r0h_system_state pre, post;
  uint32_t exit_system, exit_user;
  uint8_t input_digest[32];
  uint8_t output_digest[32];
} r0h_receipt_claim;
/* SHA-256 (FIPS 180-4) and risc0's tagged-struct digests over it */
const char* r0h_sha256(const uint8_t* bytes, size_t n, uint8_t digest_out[32]);
const char* r0h_tagged_struct(const char* tag, const uint8_t* down_digests /* n_down x 32 bytes */, size_t n_down, const uint32_t* data,
                              size_t n_data, uint8_t digest_out[32]);
const char* r0h_system_state_digest(const r0h_system_state* st, uint8_t digest_out[32]);
/* Output{journal, assumptions}.digest; assumptions_digest NULL = no assumptions (Digest::ZERO) */
const char* r0h_output_digest(const uint8_t* journal, size_t n, const uint8_t* assumptions_digest, uint8_t digest_out[32]);
const char* r0h_claim_digest(const r0h_receipt_claim* claim, uint8_t digest_out[32]);
/* the eight public-input words (canonical Montgomery words) that name a claim in a seal: Poseidon2 sponge over the sixteen 16-bit
 * halves of its digest.  The prover plants them as globals[0..8) (r0h_witgen_public / its own witness generator). */
const char* r0h_claim_globals(const uint8_t claim_digest[32], uint32_t globals_out[8]);

const char* r0h_receipt_parse(const char* json, size_t n, r0h_receipt** out);
const char* r0h_receipt_new(int kind, const uint8_t* journal, size_t journal_len, r0h_receipt** out);
const char* r0h_receipt_add_segment(r0h_receipt* rc, const uint32_t* seal, size_t seal_words, uint32_t index);
/* the same with the segment's claim (risc0-zkvm `SegmentReceipt.claim`); verifier_parameters may be NULL (zeros) */
const char* r0h_receipt_add_segment_claim(r0h_receipt* rc, const uint32_t* seal, size_t seal_words, uint32_t index,
                                          const r0h_receipt_claim* claim, const uint8_t* verifier_parameters);
const char* r0h_receipt_free(r0h_receipt* rc);
int r0h_receipt_kind(const r0h_receipt* rc);
size_t r0h_receipt_n_segments(const r0h_receipt* rc);
const char* r0h_receipt_journal(const r0h_receipt* rc, const uint8_t** bytes, size_t* n);
const char* r0h_receipt_segment(const r0h_receipt* rc, size_t i, const uint32_t** seal, size_t* seal_words, uint32_t* index);
/* *has_claim_out = 0 when the segment was read from a receipt without claims (claim_out untouched) */
const char* r0h_receipt_segment_claim(const r0h_receipt* rc, size_t i, r0h_receipt_claim* claim_out, int* has_claim_out);
const char* r0h_receipt_to_json(const r0h_receipt* rc, char** json_out);
/* `receipt.verify(image_id)` (verifier/src/main.rs:124-126, host/src/main.rs:622-624) for a composite receipt, as risc0-zkvm
 * receipt/composite.rs does it: every seal verifies against the control root of its trace size (control_roots: n_roots records of
 * 9 words [po2, root[8]], from r0h_code_root), its public inputs name the segment's claim, the segments chain (index, SystemSplit,
 * post-state == next pre-state), the last claim's output commits to SHA-256(journal.bytes) and exits Halted(0)/Paused(0), and the
 * first pre-state's digest is image_id (32 bytes; with NULL the outcome is at best R0H_RECEIPT_V_UNBOUND, never OK).  A seal of the
 * trace circuit must also carry its claim's first and last pc as public inputs 8 and 9.  Pure host code.  Returns NULL when the check
 * ran: *verdict_out is R0H_RECEIPT_V_*; *segment_out (optional) the segment at fault; *seal_verdict_out (optional) the R0H_VERIFY_*
 * code when the verdict is R0H_RECEIPT_V_SEAL. */
#define R0H_RECEIPT_V_OK 0
#define R0H_RECEIPT_V_NOT_COMPOSITE 1
#define R0H_RECEIPT_V_SEAL 2
#define R0H_RECEIPT_V_NO_CONTROL_ROOT 3
#define R0H_RECEIPT_V_NO_CLAIM 4
#define R0H_RECEIPT_V_CLAIM_MISMATCH 5
#define R0H_RECEIPT_V_CHAIN 6
#define R0H_RECEIPT_V_JOURNAL 7
#define R0H_RECEIPT_V_IMAGE_ID 8
#define R0H_RECEIPT_V_EXIT_CODE 9
#define R0H_RECEIPT_V_NO_BINDING 10
#define R0H_RECEIPT_V_HASHFN 11
#define R0H_RECEIPT_V_UNBOUND 12 /* everything else holds, but image_id was NULL: the receipt is not tied to a program */
#define R0H_RECEIPT_V_SESSION 13       /* trace circuit: a seal's session number / closing flags / challenge are not the session's */
#define R0H_RECEIPT_V_SESSION_SUM 14   /* trace circuit: the segments' sums, the program image and the journal do not balance */
#define R0H_RECEIPT_V_NEEDS_IMAGE 15   /* trace circuit: everything else holds, but the program image (the ELF) was not given */
#define R0H_RECEIPT_V_IMAGE_PROOF 16   /* r0h_receipt_verify_image: the image proof is missing, rejected, or about another image / session */
const char* r0h_receipt_verify(const r0h_receipt* rc, const uint32_t* blob, size_t blob_words, const uint32_t* control_roots,
                               size_t n_roots, const uint8_t* image_id, int* verdict_out, size_t* segment_out, int* seal_verdict_out);
/* `receipt.verify(image_id)` for receipts over the trace circuit (circuits/trace.r0c), where the program is bound by a session-wide
 * memory argument instead of in-circuit hashing: the verifier is given the ELF itself (the reference's host embeds it as
 * HYPERFRIDGE_ELF next to HYPERFRIDGE_ID, methods/build.rs:2).  Beyond what r0h_receipt_verify checks: the image id of the ELF is
 * the first pre-state; every seal carries the session challenge derived from ALL seals' DATA roots and early public inputs; segment
 * numbers count up, the closing segments are the run's last (or segments without cycles after it) with increasing address ranges;
 * and the sum of the segments' session sums equals the sum over the ELF's image words and the journal's words of their tuples'
 * fractions -- so every first touch of a word finds the image (or zero), every later one what the previous segment left, and the
 * journal is what the COMMIT rows read.  With r0h_receipt_verify (no ELF) such a receipt is at best R0H_RECEIPT_V_NEEDS_IMAGE. */
const char* r0h_receipt_verify_elf(const r0h_receipt* rc, const uint32_t* blob, size_t blob_words, const uint32_t* control_roots,
                                   size_t n_roots, const uint8_t* elf, size_t elf_len, int* verdict_out, size_t* segment_out,
                                   int* seal_verdict_out);
/* The same verdict WITHOUT the ELF: the receipt's image proof (r0h_receipt_image_proof; r0h_prove_elf attaches one when the context
 * was given the image circuit) stands for the image -- `receipt.verify(image_id)` as the reference calls it, with 32 bytes.  The image
 * seal is verified against `image_blob` bound to `image_control_root` (NULL: derived from the blob at the size the seal names). */
const char* r0h_receipt_verify_image(const r0h_receipt* rc, const uint32_t* blob, size_t blob_words, const uint32_t* control_roots,
                                     size_t n_roots, const uint32_t* image_blob, size_t image_blob_words, const uint32_t* image_control_root,
                                     const uint8_t* image_id, int* verdict_out, size_t* segment_out, int* seal_verdict_out);
const char* r0h_receipt_set_image_proof(r0h_receipt* rc, const uint32_t* seal, size_t seal_words);
const char* r0h_receipt_image_proof(const r0h_receipt* rc, const uint32_t** seal_out, size_t* seal_words_out); /* NULL / 0 when there is none */
const char* r0h_receipt_verify_reason(int verdict); /* static string, do not free */
/* The image id as text, the reference's way (host/src/main.rs:445-449, verifier/src/main.rs:131-143, host/out/IMAGE_ID.hex): eight
 * u32 words printed `{:08x}`, each stored little-endian in the 32-byte digest (`Digest::from([u32; 8])`) -- NOT the digest's bytes
 * in order.  Strict: exactly 64 hex digits. */
const char* r0h_image_id_from_hex(const char* hex, uint8_t image_id_out[32]);
const char* r0h_image_id_to_hex(const uint8_t image_id[32], char hex_out[65]);

/* ---- EBICS pre-processing: what data/checkResponse.sh does with xmllint / openssl / zlib-flate / unzip
 * before `host` starts (host/src/main.rs:143-151), as pure host code.  r0h_ebics_parse cuts the guest's four XML inputs out of a
 * response and canonicalises them exactly as the script does (checkResponse.sh:151-155, 192, 200, 221); the checks below are the
 * script's (and the guest's) checks; r0h_ebics_env_inputs frames the thirteen `ExecutorEnv` inputs (host/src/main.rs:389-417).
 * Pinned by the reference's fixtures data/test/test.xml-* (tests/test_ebics.py). ---- */
typedef struct r0h_ebics r0h_ebics;
#define R0H_EBICS_AUTHENTICATED 0       /* "<xml>-authenticated": header, DataEncryptionInfo, ReturnCode, [TimestampBankParameter] */
#define R0H_EBICS_SIGNED_INFO 1         /* "<xml>-SignedInfo" */
#define R0H_EBICS_SIGNATURE_VALUE 2     /* "<xml>-SignatureValue" (element with its tags) */
#define R0H_EBICS_ORDER_DATA 3          /* "<xml>-OrderData" (element with its tags) */
#define R0H_EBICS_DIGEST_VALUE 4        /* text of <ds:DigestValue> */
#define R0H_EBICS_SIGNATURE_BIN 5       /* base64-decoded signature */
#define R0H_EBICS_TRANSACTION_KEY_BIN 6 /* base64-decoded <TransactionKey> (the RSA ciphertext) */
#define R0H_EBICS_ORDER_DATA_BIN 7      /* base64-decoded order data (AES ciphertext) */
#define R0H_EBICS_PAYLOAD_ZIP 8         /* decrypted and inflated payload (after r0h_ebics_decrypt_order_data) */
const char* r0h_ebics_parse(const char* xml, size_t n, r0h_ebics** out);
const char* r0h_ebics_free(r0h_ebics* e);
const char* r0h_ebics_part(const r0h_ebics* e, int which, const uint8_t** bytes, size_t* n);
/* each check returns NULL when it ran; *ok_out = 1 passed / 0 failed */
const char* r0h_ebics_check_digest(const r0h_ebics* e, int* ok_out);
const char* r0h_ebics_verify_bank_signature(const r0h_ebics* e, const char* pub_bank_pem, size_t pem_len, int* ok_out);
/* raw_block: the RSA-decrypted transaction key with its padding ("<xml>-TransactionKeyDecrypt.bin"); key_out: the AES key in it */
const char* r0h_ebics_check_transaction_key(const r0h_ebics* e, const char* pub_client_pem, size_t pem_len, const uint8_t* raw_block,
                                            size_t raw_len, uint8_t key_out[16], int* ok_out);
/* the two private-key steps of the script (checkResponse.sh:231-236, 276-279; `openssl pkeyutl -decrypt / -sign`): the raw
 * RSA-decrypted transaction key block ("<xml>-TransactionKeyDecrypt.bin") with the AES key in it, and the witness signature as the
 * `xxd -p` text of "<xml>-Witness.hex" (free with r0h_free_error).  PKCS#8 or PKCS#1 PEM; plain square-and-multiply, not constant
 * time: a tool for the key's owner, as the openssl command line in the script is. */
const char* r0h_ebics_decrypt_transaction_key(const r0h_ebics* e, const char* client_private_pem, size_t pem_len, uint8_t* raw_out,
                                              size_t raw_capacity, size_t* raw_len_out, uint8_t key_out[16], int* ok_out);
const char* r0h_ebics_witness_sign(const r0h_ebics* e, const char* witness_private_pem, size_t pem_len, char** hex_out);
const char* r0h_ebics_verify_witness(const r0h_ebics* e, const char* pub_witness_pem, size_t pem_len, const char* witness_hex,
                                     size_t hex_len, int* ok_out);
/* AES-128-CBC (zero IV) -> RFC 1950 inflate -> ZIP members; an error means the key or the data is wrong */
const char* r0h_ebics_decrypt_order_data(r0h_ebics* e, const uint8_t key[16]);
size_t r0h_ebics_n_documents(const r0h_ebics* e);
const char* r0h_ebics_document(const r0h_ebics* e, size_t i, const char** name, const uint8_t** data, size_t* n);
/* modulus and exponent of a PEM "PUBLIC KEY" as decimal strings (host/src/main.rs:383-387); free both with r0h_free_error */
const char* r0h_rsa_public_key_decimal(const char* pem, size_t pem_len, char** modulus_out, char** exponent_out);
const char* r0h_ebics_env_inputs(const r0h_ebics* e, const char* pub_bank_pem, size_t bank_len, const char* client_private_pem,
                                 size_t client_len, const uint8_t* decrypted_tx_key, size_t tx_len, const char* iban,
                                 const char* host_info, const char* witness_hex, size_t witness_len, const char* pub_witness_pem,
                                 size_t pub_witness_len, const char* verbose, r0h_env** out);
/* known-answer hooks: one AES-128 block (FIPS 197), one RFC 1950 stream (caller frees *out with r0h_free_error) */
/* The input word stream of this library's own camt53 guest (tools/guest_camt53.py, circuits/guest_camt53.elf) from the same things the
 * reference's host gives its guest (host/src/main.rs:389-417): the parsed response, the three public keys, the decrypted transaction
 * key block (256 bytes: r0h_ebics_decrypt_transaction_key or the script's TransactionKeyDecrypt.bin), the witness signature (hex),
 * iban and host info, the commitment form (1: with the three keys, as the current receipts; 0: the earlier form).  words_out is
 * malloc'd: release it with r0h_free_error (as every buffer this library hands out).  With it the compiled hosts go from response.xml
 * to a receipt without Python. */
const char* r0h_camt53_guest_input(const r0h_ebics* e, const char* pub_bank_pem, size_t bank_len, const char* pub_client_pem,
                                   size_t client_len, const char* pub_witness_pem, size_t witness_len, const uint8_t* tx_key_block,
                                   size_t tx_key_len, const char* witness_hex, size_t witness_hex_len, const char* iban,
                                   const char* host_info, uint32_t form, uint32_t** words_out, size_t* n_out);
const char* r0h_aes128_block(const uint8_t key[16], const uint8_t in[16], int decrypt, uint8_t out[16]);
const char* r0h_zlib_inflate(const uint8_t* in, size_t n, uint8_t** out, size_t* out_len);

/* ---- RV32IM executor, segmenter and preflight trace (csrc/rv32im.cpp): the part of `prover.prove(env, elf)`
 * that runs before prove_segment.  Pure host code.  Instruction semantics are the RISC-V specification's; the ecall ABI, the cycle
 * model and the page Merkle root are this library's own documented choices (risc0's are recalled in outline only: see the source).
 * Guest memory is the low 1 GiB (an access or a pc above it traps).
 * ecall (a7): 0 HALT(a0) | 1 READ_WORDS(a0 = dst, a1 = n words) | 2 COMMIT(a0 = src, a1 = n words) | 3 CYCLES -> a0 | 4 PAUSE(a0);
 * every ecall cycle reads a7 and a0 (its two register reads).  The two transfers move ONE word per cycle and keep no state
 * outside the registers: while a1 = j > 0 the instruction re-executes (next pc = pc) -- it moves word j - 1 of the buffer
 * (address a0 + 4 (j - 1)) and writes a1 = j - 1 -- and with a1 = 0 it falls through to pc + 4 (n + 1 cycles for n words).  So
 * every cycle has at most one memory access and is a function of what it reads -- what the trace circuit constrains -- and a
 * transfer of any length can be cut between two segments.  The buffer ends up in stream order; a1 is 0 afterwards, a0 as it was;
 * a0 is word-aligned. ---- */
typedef struct r0h_vm r0h_vm;
typedef struct {
  uint32_t segment_po2;     /* a segment holds at most 2^segment_po2 rows (cycles + paging + boundary rows) */
  uint32_t page_in_cycles;  /* charged once per 1 KiB page first touched in a segment */
  uint32_t page_out_cycles; /* charged once per page written in a segment */
  uint32_t keep_trace;      /* record one r0h_preflight_row per cycle and the boundary rows of every segment */
  uint64_t max_cycles;      /* stop (R0H_VM_LIMIT, ExitCode::SessionLimit) after this many cycles; 0 = no limit */
  uint32_t boundary_rows;   /* charge one row per distinct register / memory word touched in a segment: the rows the trace
                             * circuit spends on the first and last value of each (set by r0h_prove_elf for a trace circuit) */
  uint32_t reserved;
} r0h_vm_limits;
typedef struct {
  uint32_t index, exit_system, exit_user, pages_in, pages_out, boundary_rows;
  uint32_t closing, reserved; /* closing = 1: the boundary rows of this segment close the session (see r0h_preflight_bound) */
  uint64_t user_cycles, paging_cycles;
  r0h_system_state pre, post; /* pc + Merkle root of memory before the first and after the last instruction of the segment */
} r0h_vm_segment;
#define R0H_MEM_NONE 0
#define R0H_MEM_READ 1
#define R0H_MEM_WRITE 2
/* What witness generation replays, one row per cycle (18 words): the instruction, its register operands and result, its memory
 * transaction, and -- for the memory-consistency argument of the trace circuit -- when each of the five things the cycle touches
 * was last touched in this segment.  Access k of cycle c carries the timestamp 5 c + k + 1 (k = 0 x[rs1] read, 1 x[rs2] read,
 * 2 x[rd] write, 3 the memory word, 4 the instruction fetch); prev[k] is the timestamp of the previous access to the same
 * register / word in this segment, 0 when this is the first.  rs1 / rs2 are instruction bits 15..19 / 20..24 whatever the format. */
typedef struct {
  uint32_t cycle;  /* within the segment */
  uint32_t pc, insn, next_pc;
  uint32_t rs1_value, rs2_value;
  uint32_t rd, rd_before, rd_after;                    /* rd = 0: no register written */
  uint32_t mem_kind, mem_addr, mem_before, mem_after; /* word-aligned address, the word before and after */
  uint32_t prev[5];
} r0h_preflight_row;
/* One per register / memory word a segment touched, in increasing address order: the value the segment found there, the value it
 * left, and the timestamp of its last access.  addr: word index (byte address / 4) for memory, R0H_REG_BASE + i for x[i]. */
#define R0H_REG_BASE 0x10000000u
#define R0H_BOUND_IMAGE 1u
/* prev_seg: the number (index + 1) of the segment that last touched the address before this one, 0 if none did.  In a CLOSING
 * segment (the run's last, or segments of boundary rows only that follow it when those do not fit) there is a row for every word
 * and register the session touched and for every word of the program image, touched or not: init_value is what the address held
 * when the run began (the image's word, or zero) and flags says whether it is an image word. */
typedef struct { uint32_t addr, first_value, last_value, last_ts, prev_seg, init_value, flags, reserved; } r0h_preflight_bound;
/* the journal is a window of guest memory: journal word i is the word at R0H_JOURNAL_BASE + 4 i when COMMIT names it */
#define R0H_JOURNAL_BASE 0x20000000u
#define R0H_VM_HALTED 0
#define R0H_VM_PAUSED 1
#define R0H_VM_LIMIT 2
const char* r0h_vm_new(r0h_vm** out);
const char* r0h_vm_free(r0h_vm* vm);
const char* r0h_vm_load(r0h_vm* vm, uint32_t addr, const uint32_t* words, size_t n);
const char* r0h_vm_load_elf(r0h_vm* vm, const uint8_t* elf, size_t n); /* ELF32 LE RISC-V executable: PT_LOAD segments + entry */
/* the image id of an ELF -- risc0-binfmt `compute_image_id`, what `methods/build.rs` embeds as HYPERFRIDGE_ID: the digest of the
 * SystemState a run starts from (r0h_prove_elf returns the same 32 bytes).  Pure host code. */
const char* r0h_compute_image_id(const uint8_t* elf, size_t n, uint8_t image_id_out[32]);
const char* r0h_vm_set_input(r0h_vm* vm, const uint32_t* words, size_t n); /* the ExecutorEnv word stream (r0h_env_words) */
const char* r0h_vm_set_pc(r0h_vm* vm, uint32_t pc);
const char* r0h_vm_set_reg(r0h_vm* vm, uint32_t i, uint32_t value);
uint32_t r0h_vm_reg(const r0h_vm* vm, uint32_t i);
uint32_t r0h_vm_pc(const r0h_vm* vm);
const char* r0h_vm_read(const r0h_vm* vm, uint32_t addr, uint32_t* words, size_t n);
/* runs to HALT / PAUSE / max_cycles, cutting segments on the way; a guest trap (illegal instruction, misaligned access, unknown
 * ecall) is an error string, as a guest panic is an Err from `prove` (host/src/main.rs:327-330) */
const char* r0h_vm_run(r0h_vm* vm, const r0h_vm_limits* limits, int* exit_kind_out, uint32_t* exit_code_out);
/* the same run one segment at a time (a prover takes each segment as it is cut, while the guest runs on): returns after the next
 * segment is complete; *finished_out = 1 with the last one (then exit_kind / exit_code are set).  The limits of the first call hold
 * for the whole run.  r0h_vm_release_trace drops the rows of a segment that has been taken (its r0h_vm_segment stays). */
const char* r0h_vm_run_segment(r0h_vm* vm, const r0h_vm_limits* limits, int* finished_out, int* exit_kind_out, uint32_t* exit_code_out);
const char* r0h_vm_release_trace(r0h_vm* vm, size_t i);
size_t r0h_vm_n_segments(const r0h_vm* vm);
uint64_t r0h_vm_cycles(const r0h_vm* vm);
const char* r0h_vm_segment_info(const r0h_vm* vm, size_t i, r0h_vm_segment* out);
const char* r0h_vm_preflight(const r0h_vm* vm, size_t i, const r0h_preflight_row** rows, size_t* n);
const char* r0h_vm_boundary(const r0h_vm* vm, size_t i, const r0h_preflight_bound** rows, size_t* n);
/* ---- the trace circuit, version 5 (circuits/trace.r0c, tools/trace_circuit.py): a circuit whose DATA group IS the preflight trace of a
 * segment.  R0H_TRACE_COLUMNS columns of 2^po2 rows (po2 >= 16: two 2^16-row lookup tables sit in the CODE group), column-major,
 * Montgomery words: first the cycles (one row each), then the boundary rows (one per register / word touched; in a closing segment
 * one per word the whole session touched and per image word), then blank rows.  Per cycle: pc, next pc, the instruction word as
 * decoded fields with a one-hot opcode and funct3, five accesses -- the fetch, x[rs1], x[rs2], x[rd], the memory word, in this order of
 * timestamps (5 c + 1..5) -- each with the two looked-up limbs of the distance to the access it follows, and the work words of the
 * arithmetic units: two operands byte by byte with their AND (U, V), two words as 16-bit halves (Z, W), carries.  Ten quantities the
 * constraints speak of are linear forms of these columns rather than columns (the last flag of each one-hot group, V's low byte, Z's
 * low half, the memory and transfer flags, four of the five consumed timestamps): 128 columns, eight Poseidon2 permutations per
 * committed row.  What the circuit constrains (tools/trace_circuit.py trace_constraints: 302 polynomials of degree <= 5 over 193
 * taps; csrc/trace.hpp fills the columns):
 *   - the cycles form one contiguous run from the public first pc to the public last pc in the public number of cycles;
 *   - WHAT EVERY INSTRUCTION DOES: the word decodes to exactly one RV32IM instruction (an illegal encoding has no satisfying row);
 *     the value written to rd, the word written to memory, the address of a load / store and the next pc are the ones the ISA
 *     prescribes for the operands read -- a 32-bit adder over 16-bit halves (ADD[I], AUIPC, addresses, JALR; backwards for SUB,
 *     SLT[I][U] and the branches), AND / OR / XOR through a byte-AND lookup table, a byte-limb multiplier with a range-checked carry
 *     chain (MUL[H[S]U], the shifts as products with 2^s or 2^(32-s), DIV[U] / REM[U] as quotient x divisor + remainder = dividend
 *     with |remainder| < |divisor|), byte and half lanes of the narrow loads and stores; an ecall reads a7 and a0, and what it writes
 *     (a1 counted down and one word of the buffer for the transfers, a0 for CYCLES, nothing for HALT / PAUSE) is where its
 *     function says.  Every word written to a register or to memory is range-checked (16-bit lookups) or composed of such parts;
 *   - MEMORY CONSISTENCY over registers (x0 included: a register nothing ever writes) and memory: every access consumes the tuple
 *     (address, value, timestamp, space) the previous access to that address produced and produces one with a larger timestamp; the
 *     boundary rows produce each address's first tuple (timestamp 0) and consume its last; consumed = produced as multisets, by a
 *     log-derivative argument: every tuple is a fraction +-1 / (alpha - address - b1 lo - b2 hi - b3 t - b4 space) of ONE running
 *     sum through the ACCUM group that wraps around the trace, together with the lookups' fractions and the tables' multiplicities
 *     (csrc/logup.hip).  Boundary addresses (below 2^28 + 32) strictly increase, so an address has ONE history.  Hence a register
 *     or word read returns what was last written to it, and the instruction word at a pc is the word in memory;
 *   - THE SESSION (round 4): a boundary row consumes (address, value found, number of the segment that held the address before --
 *     an EARLIER one) and produces (address, value left, own number); the rows of a closing segment produce the address's initial
 *     tuple (address, initial value, 0) instead -- zero unless the row says "image word", then it also names (address, word) as an
 *     image tuple; an active COMMIT row names (address, word read) as a journal tuple.  These are fractions under a challenge all
 *     segments share (late public inputs R0H_TRACE_GAMMA..: derived from every segment's DATA root, r0h_session_challenge); their
 *     sum over the segment is the public input at R0H_TRACE_SUM.  r0h_receipt_verify_elf adds the segments' sums and compares with
 *     the sum over the ELF's image words and the journal's words: the tuples balance iff every first touch finds the image (or
 *     zero), every later one what the previous holder left, and the journal is what the COMMIT rows read.
 * What it does NOT constrain: the WORDS READ_WORDS moves (input words are the host's to choose, as in risc0) and the value CYCLES
 * returns.  It is this library's circuit for this library's executor, not risc0's rv32im circuit.  Public inputs
 * (R0H_TRACE_GLOBALS): 8 words naming the segment's ReceiptClaim (r0h_claim_globals), first pc, pc after the last cycle, number of
 * cycles, how the segment ends (0: cut, 1: HALT, 2: PAUSE -- such an ecall is the last cycle of its segment), that being non-zero,
 * the two halves of the exit code (a0), the segment's number (index + 1), whether it closes the session, number x (1 - closing), the
 * first and last boundary address; LATE (absorbed after the DATA commitment): the session challenge (16 words), the segment's sum
 * (4).  r0h_receipt_verify[_elf] holds them against the claims and against one another.  Trace sizes 2^16..2^21 rows (timestamps < 2^24). ---- */
#define R0H_TRACE_COLUMNS 128
#define R0H_TRACE_GLOBALS 40
#define R0H_TRACE_LATE_GLOBALS 20 /* the last 20 public inputs: the session's challenge (16 words) and the segment's sum under it (4) */
#define R0H_TRACE_GAMMA 20         /* where the session challenge sits among the public inputs; the segment's sum follows at 36 */
#define R0H_TRACE_SUM 36
#define R0H_SESSION_TAG_IMAGE ((1u << 20) + 1)   /* the "segment" coordinate of an image word's tuple in the session sum */
#define R0H_SESSION_TAG_JOURNAL ((1u << 20) + 2) /* ... and of a journal word's */
#define R0H_SESSION_RECORD_WORDS 28 /* what a segment contributes to the session challenge: its 20 early public inputs, its DATA root */
#define R0H_TRACE_MIN_PO2 16      /* the lookup tables have 2^16 rows */
#define R0H_TRACE_MAX_PO2 21
const char* r0h_trace_column_name(uint32_t column); /* static string; NULL past the last column */
/* host reference of the witness (what tests compare the device kernel with): data_out = R0H_TRACE_COLUMNS * 2^po2 words;
 * globals_out[8..15) = first pc, pc after the last cycle, cycles, how the segment ends, that being non-zero, exit code halves (globals_out[0..8) are left to the caller: the claim) */
const char* r0h_vm_trace_witness(const r0h_vm* vm, size_t i, uint32_t po2, uint32_t* data_out, uint32_t globals_out[R0H_TRACE_GLOBALS]);
/* the same on the device: the compact rows (72 B per cycle, 16 B per boundary row) are uploaded and one thread per row expands
 * them into the column-major Montgomery DATA group in `data` (R0H_TRACE_COLUMNS * 2^po2 words).  Stream-ordered; the host arrays
 * may be released when the call returns. */
typedef struct { uint32_t number /* index + 1 */, closing, idle_pc /* where a segment without cycles stands */, reserved; } r0h_trace_segment;
const char* r0h_trace_witgen(r0h_ctx* ctx, const r0h_preflight_row* rows, size_t n_rows, const r0h_preflight_bound* bounds,
                             size_t n_bounds, uint32_t po2, const r0h_trace_segment* segment, r0h_buf* data,
                             uint32_t globals_out[R0H_TRACE_GLOBALS]);
/* ---- the log-derivative argument of a circuit (blob section LOGUP): lookups, memory tuples, session tuples as fractions of running
 * sums in ACCUM.  Multiplicities: the table columns of DATA are filled from the lookups the rows make -- BEFORE the DATA group is
 * committed (r0h_trace_witgen and r0h_vm_trace_witness leave them zero).  Totals: the accumulators whose challenges are public inputs
 * (the trace circuit's session sum) are summed over the rows and written into global_io where the circuit reads them -- once those
 * challenges are known, before r0h_proof_late.  r0h_accum_public is r0h_accum for circuits whose accumulation reads public inputs. */
const char* r0h_logup_multiplicities(r0h_ctx* ctx, const r0h_circuit* c, uint32_t po2, r0h_buf* data, const uint32_t* global);
const char* r0h_logup_multiplicities_host(const uint32_t* blob, size_t blob_words, uint32_t po2, uint32_t* data, const uint32_t* global);
const char* r0h_logup_totals(r0h_ctx* ctx, const r0h_circuit* c, uint32_t po2, const r0h_buf* code, const r0h_buf* data, uint32_t* global_io);
const char* r0h_accum_public(r0h_ctx* ctx, const r0h_circuit* c, uint32_t po2, const r0h_buf* code, const r0h_buf* data,
                             const uint32_t* global, const uint32_t* mix, r0h_buf* accum);
const char* r0h_vm_journal(const r0h_vm* vm, const uint8_t** bytes, size_t* n);
/* the ReceiptClaim of segment i: system states and exit code from the run, Output{journal} on the last segment */
const char* r0h_vm_segment_claim(const r0h_vm* vm, size_t i, r0h_receipt_claim* out);

/* ---- `default_prover().prove(env, elf)` (host/src/main.rs:420-423) in one call: execute the ELF on the word stream of `env`,
 * cut the run into segments of at most 2^segment_po2 rows, prove each at the smallest trace size that holds it, return the
 * composite receipt with every segment's claim bound to its seal, plus the image id `receipt.verify` is given.  A guest that
 * traps, exits non-zero or exceeds max_cycles is an error, as it is an Err from `prove`; max_cycles = 0 selects
 * R0H_DEFAULT_SESSION_LIMIT (risc0's executor has a session limit as well).
 * With the trace circuit (circuits/trace.r0c) every seal attests the segment it stands for: the executor keeps the preflight
 * rows, r0h_trace_witgen expands them on the device, the public inputs carry the claim, the first and last pc and the cycle count,
 * and the guest runs ahead on its own thread while the device proves.  With any other circuit the witness is that circuit's
 * synthetic column program with the claim planted (the seal then proves only that a satisfying trace naming the claim exists). ---- */
#define R0H_DEFAULT_SESSION_LIMIT ((uint64_t)1 << 32)
/* How many bytes of committed DATA evaluations the sessions begun on this context keep on the device between their two phases; the
 * segments beyond it keep coefficients and Merkle nodes only (0.6 instead of 2.7 GiB per 2^20-row segment) and are evaluated again
 * when their proofs are finished (about 2 ms each).  0 = the default: an eighth of the device's memory.  The reference's own run is
 * 37 segments (docs/runtime.md: session_cycles = 39,265,237): 100 GiB resident without a limit, per session in flight. */
const char* r0h_ctx_set_session_resident_limit(r0h_ctx* ctx, uint64_t bytes);
const char* r0h_prove_elf(r0h_ctx* ctx, const r0h_circuit* c, const uint8_t* elf, size_t elf_len, const uint32_t* input_words,
                          size_t n_input, uint32_t segment_po2, uint64_t max_cycles, r0h_receipt** receipt_out,
                          uint8_t image_id_out[32], uint64_t* cycles_out);
/* One rank's share of a session that is proved on several GPUs: the guest is executed in full on every rank (it is deterministic and
 * costs a tenth of a second per ten million cycles -- less than moving 72 MiB of rows per segment between GPUs), segments part,
 * part + parts, ... are proved.  The receipt holds those segments only; r0h_receipt_merge puts the ranks' receipts together into the
 * receipt r0h_prove_elf would have returned (same seals: a segment's proof does not depend on who makes it).  No collective is
 * involved: the receipts travel as JSON (hyperfridge-r0_amd/driver.py: prove_elf_sharded over torch.distributed). */
const char* r0h_prove_elf_part(r0h_ctx* ctx, const r0h_circuit* c, const uint8_t* elf, size_t elf_len, const uint32_t* input_words,
                               size_t n_input, uint32_t segment_po2, uint64_t max_cycles, uint32_t part, uint32_t parts,
                               r0h_receipt** receipt_out, uint8_t image_id_out[32], uint64_t* cycles_out);
/* The two phases of a trace-circuit session, for ranks that share one (r0h_prove_elf is begin + finish on one rank): `begin` executes
 * the guest and commits the DATA group of this rank's segments (part, part + parts, ...); `records` gives what they contribute to the
 * session challenge; the ranks exchange their records (28 words per segment: an all-gather); `finish` takes the records of ALL
 * segments in index order, derives the challenge, finishes this rank's proofs and returns their receipt (r0h_receipt_merge joins
 * the ranks' receipts).  A session holds device memory (about 3 GiB per 2^20-row segment) until it is finished or freed. */
/* the challenge itself (public inputs R0H_TRACE_GAMMA .. +16 of every seal of the session): alpha_g, gamma, gamma^2, gamma^3 drawn
 * from the Poseidon2 digest of all records in index order -- what r0h_session_finish and r0h_receipt_verify[_elf] both compute */
const char* r0h_session_challenge(const uint32_t* records, size_t n_records, uint32_t challenge_out[16]);
typedef struct r0h_session r0h_session;
const char* r0h_session_begin(r0h_ctx* ctx, const r0h_circuit* c, const uint8_t* elf, size_t elf_len, const uint32_t* input_words,
                              size_t n_input, uint32_t segment_po2, uint64_t max_cycles, uint32_t part, uint32_t parts,
                              r0h_session** session_out);
size_t r0h_session_n_segments(const r0h_session* s); /* of the whole session */
const char* r0h_session_records(const r0h_session* s, uint32_t* indices_out, uint32_t* records_out, size_t capacity, size_t* n_out);
const char* r0h_session_finish(r0h_session* s, const uint32_t* all_records, size_t n_records, r0h_receipt** receipt_out,
                               uint8_t image_id_out[32], uint64_t* cycles_out);
const char* r0h_session_free(r0h_session* s);
/* composite receipts that each hold some segments of one session -> one receipt with all of them in index order.  Refused: a
 * segment index missing or present twice, journals that differ, a receipt that is not composite. */
const char* r0h_receipt_merge(const r0h_receipt* const* parts, size_t n, r0h_receipt** out);
/* per-stage timing of the last r0h_prove_elf on this context (for tools/bench_session.py): names are static strings */
typedef struct { uint32_t segments, lean_segments /* proved without resident evaluations: r0h_ctx_set_session_resident_limit */; uint64_t cycles; double executor_s, witgen_ms, prove_ms, wall_s; } r0h_session_stats;
const char* r0h_last_session_stats(r0h_ctx* ctx, r0h_session_stats* out);

/* ---- the image proof: `receipt.verify(image_id)` without the program image (verifier/src/main.rs:124-126).  r0h_receipt_verify_elf
 * completes a trace-circuit session's memory argument with the ELF's own words; a verifier who holds only the 32 bytes of the image id
 * cannot.  The image circuit (circuits/image.r0c, tools/image_circuit.py; W = 8 ACCUM + 30 CODE + 69 DATA) proves the image's side
 * for it: its rows run the Poseidon2 sponge over the image's word list -- blocks of four (word index, low half, high half) and a mask
 * -- whose digest is the root of the initial memory state the image id names (public inputs 0..7), and add every word's fraction
 * 1 / (alpha_g - index - gamma lo - gamma^2 hi - gamma^3 TAG_IMAGE) under the session's challenge (inputs 8..23) to a running sum
 * whose total is public (inputs 24..27).  r0h_prove_elf attaches such a seal to the receipt when the context has been given the image
 * circuit (r0h_ctx_set_image_circuit); r0h_receipt_verify_image checks it and balances the session with its total. ---- */
#define R0H_IMAGE_COLUMNS 69
#define R0H_IMAGE_GLOBALS 28
#define R0H_IMAGE_LATE_GLOBALS 20
#define R0H_IMAGE_GAMMA 8
#define R0H_IMAGE_SUM 24
/* smallest trace that holds the image's sponge (30 rows per four words; at least 2^9) */
const char* r0h_image_po2(const uint8_t* elf, size_t elf_len, uint32_t* po2_out);
/* host: the image circuit's DATA group for an ELF ([R0H_IMAGE_COLUMNS][2^po2], Montgomery, column-major) and its public inputs with
 * the digest filled in (challenge and total left zero) */
const char* r0h_image_witness(const uint8_t* elf, size_t elf_len, uint32_t po2, uint32_t* data_out, uint32_t globals_out[R0H_IMAGE_GLOBALS]);
/* sessions begun on `ctx` afterwards (r0h_prove_elf, r0h_session_begin with part 0) attach an image proof to their receipts; NULL stops
 * that.  The circuit must have been loaded on this context and must outlive the sessions. */
const char* r0h_ctx_set_image_circuit(r0h_ctx* ctx, const r0h_circuit* image_circuit);
/* device: one image proof under `challenge` (r0h_session_challenge); seal_out may be NULL to ask for the size */
const char* r0h_prove_image(r0h_ctx* ctx, const r0h_circuit* image_circuit, const uint8_t* elf, size_t elf_len, const uint32_t challenge[16],
                            uint32_t* seal_out, size_t seal_capacity_words, size_t* seal_words_out);

/* ---- recursion: risc0-zkvm `ProverServer::{lift, join}` (risc0-circuit-recursion 4.0.4, Cargo.lock:3050-3085; BASELINE.json
 * configs[4]).  `lift` stands one recursion-circuit proof for one segment seal and its claim; `join` folds two nodes into one whose
 * claim is the composition {pre: a.pre, post: b.post, exit_code: b.exit_code, input: a.input, output: b.output}, and refuses two
 * nodes that do not follow one another (a must end in SystemSplit with a.post == b.pre).  A node's 16 public inputs are the 8 words
 * naming its composed claim (r0h_claim_globals) and the Poseidon2 digest of what it consumed -- the segment seal's words for a lift,
 * the two child seals' digests for a join.  That digest is computed INSIDE the node's proof: the circuit's sponge component runs the
 * permutation one round per row over witness cells holding the consumed words and ties the result to public inputs 8..15, so a
 * witness holding other words than the digest names satisfies no trace.  A lift of a 2^20-row trace-circuit seal (61k words: 3.8k
 * permutations of 30 rows) needs a recursion trace of 2^17 rows or more.
 * NOT risc0's recursion circuit: that in-circuit hash is the first and only in-circuit step.  Every node is a proof over this
 * repository's recursion circuit (circuits/recursion.r0c) made with the same kernels, and the SEALS it consumes are verified BESIDE
 * that proof (host threads, while the device proves), not inside it -- a root is a checkable tree of seals carrying the end-to-end
 * claim, not a succinct receipt.  Moving nodes between ranks
 * (the tree's levels) is the caller's: hyperfridge-r0_amd/recursion.py does it over torch.distributed point-to-point. ---- */
typedef struct r0h_recursor r0h_recursor;
typedef struct r0h_node r0h_node;
/* segment_control_roots: n_roots records of 9 words [po2, root[8]] (r0h_code_root of the segment circuit); with n_roots = 0 the
 * leaves are verified against the segment circuit alone */
const char* r0h_recursor_new(r0h_ctx* ctx, const uint32_t* recursion_blob, size_t recursion_words, const char* code_object_path,
                             uint32_t po2, const uint32_t* segment_blob, size_t segment_words, const uint32_t* segment_control_roots,
                             size_t n_roots, r0h_recursor** out);
const char* r0h_recursor_free(r0h_recursor* rc);
const char* r0h_recursor_control_root(const r0h_recursor* rc, uint32_t root_out[8]); /* of the recursion circuit at its po2 */
const char* r0h_lift(r0h_recursor* rc, const uint32_t* seal, size_t seal_words, const r0h_receipt_claim* claim, r0h_node** out);
const char* r0h_join(r0h_recursor* rc, const r0h_node* a, const r0h_node* b, r0h_node** out);
const char* r0h_node_new(const uint32_t* seal, size_t seal_words, const r0h_receipt_claim* claim, r0h_node** out); /* as received */
const char* r0h_node_free(r0h_node* node);
const char* r0h_node_seal(const r0h_node* node, const uint32_t** seal, size_t* seal_words);
const char* r0h_node_claim(const r0h_node* node, r0h_receipt_claim* claim_out);
const char* r0h_node_verify(const uint32_t* recursion_blob, size_t blob_words, const uint32_t* control_root, const r0h_node* node,
                            int* ok_out);

/* Optional per-kernel timing with HIP events on the context's stream (for bench.py's roofline object): enable, run,
 * then read {"kernel family": {"launches", "total_ms", "alg_bytes"}} as JSON.  Enabling resets the counters. */
const char* r0h_kernel_timing(r0h_ctx* ctx, int enable);
const char* r0h_kernel_stats(r0h_ctx* ctx, char* json_out, size_t capacity);

#ifdef __cplusplus
}
#endif
#endif
