// Parsed circuit blob (include/r0hip_circuit.h) shared by circuit.hip (loading, eval_check) and prover.hip (sequencer).
#pragma once
#include "internal.hpp"

namespace r0h {

struct Tap { uint32_t group, offset, back; };
struct Reg { uint32_t group, offset, first_tap, size, combo; };
struct Step { uint32_t op, a, b, c; };
struct CodeCol { uint32_t kind, param; };
struct DataCol { uint32_t kind, a, b, c, e; };
struct AccCol { uint32_t first, a, b; };
struct AccFp { uint32_t n_f; uint32_t col[3][4]; };  // running product of up to three tuple fingerprints (R0H_SEC_ACCUM_FP)
struct Term { uint32_t pow, v; std::vector<uint32_t> conds; };
// the log-derivative argument (R0H_SEC_LOGUP): fractions numerator / (sum of challenge x linear form), four to an accumulator
struct LfTerm { uint32_t coef, global, col; };  // canonical coefficient; public input + 1 or 0; column ref + 1 or 0 (the constant one)
struct Lf { std::vector<LfTerm> terms; };
struct LogupPart { uint32_t ch_kind, ch_idx; Lf lf; };
struct LogupFraction { uint32_t table; Lf num; std::vector<LogupPart> parts; };
struct LogupAcc { uint32_t final_global; std::vector<LogupFraction> fr; };
struct LogupTable { uint32_t data_col, kind; };
struct Logup {
  std::vector<LogupTable> tables;
  std::vector<LogupAcc> accs;
  uint32_t n_chain = 0;  // the first n_chain accumulators are links of the chain
};

struct Plan {                      // how the constraint program is cut into kernels
  std::vector<Term> terms;         // flattened, in chain order
  std::vector<uint32_t> cut;       // kernel k owns terms [cut[k], cut[k+1])
  uint32_t n_pow = 0;
};

}  // namespace r0h

struct r0h_circuit {
  r0h_ctx* ctx = nullptr;
  uint32_t group_size[3] = {0, 0, 0};
  std::vector<r0h::Tap> taps;
  std::vector<r0h::Reg> regs;
  std::vector<uint32_t> combo_begin, combo_backs;
  uint32_t group_tap_begin[4] = {0, 0, 0, 0};
  uint32_t n_global = 0, n_mix = 0;
  std::vector<uint32_t> global_cols;
  std::vector<r0h::Step> steps;
  uint32_t ret = 0;
  std::vector<uint32_t> fp_step, mix_step;  // variable index -> step index
  std::vector<r0h::CodeCol> code_cols;
  std::vector<r0h::DataCol> data_cols;
  std::vector<r0h::AccCol> acc_cols;
  std::vector<r0h::AccFp> acc_fp;
  r0h::Logup logup;
  uint32_t period = 0;               // R0H_SEC_PERIODIC: what CODE columns of kind 6 repeat, [n_periodic][period] canonical values
  std::vector<uint32_t> periodic;
  bool has_sponge = false;           // R0H_SEC_SPONGE: the in-circuit Poseidon2 sponge, its first CODE / DATA column, the first public input of its digest
  uint32_t sponge_code = 0, sponge_data = 0, sponge_global = 0;
  uint32_t n_late = 0;  // the last n_late public inputs enter the transcript after the DATA commitment (R0H_SEC_LATE)
  bool has_column_program = false;  // WITGEN + ACCUM present (synthetic circuits); imported circuits bring their own witness
  std::vector<uint32_t> blob;
  uint8_t info[16] = {'R', '0', 'H', 'I', 'P', '_', 'S', 'Y', 'N', 'T', 'H', ':', 'v', '1', '_', '_'};  // circuit ProtocolInfo tag
  r0h::Plan plan;
  hipModule_t module = nullptr;
  std::vector<hipFunction_t> kernels;
};

namespace r0h {
// fills the host tables of `c` from a blob (no device work); validates every index the sequencer and the verifier follow
const char* parse_blob(r0h_circuit* c, const uint32_t* blob, size_t n_words);
// the log-derivative accumulation on the device (logup.hip): multiplicities into DATA, the ACCUM group, totals of the public accumulators
// the rows of the in-circuit sponge over `words` written into the circuit's sponge columns of `data` (recursion.cpp: sponge_plant)
const char* sponge_plant(r0h_ctx* ctx, const r0h_circuit* c, uint32_t po2, const uint32_t* words, size_t n_words, r0h_buf* data);
const char* logup_accum(r0h_ctx* ctx, const r0h_circuit* c, uint32_t po2, const r0h_buf* code, const r0h_buf* data, const uint32_t* global, const uint32_t* mix, r0h_buf* accum);
}  // namespace r0h
