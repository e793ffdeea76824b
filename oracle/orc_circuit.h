/* ORACLE -- TEST INFRASTRUCTURE ONLY.  Internal view of a parsed circuit blob (format: include/r0hip_circuit.h). */
#ifndef ORC_CIRCUIT_H
#define ORC_CIRCUIT_H
#include "orc.h"
#include "orc_field.h"

typedef struct { uint32_t group, offset, back; } orc_tap_t;
typedef struct { uint32_t group, offset, first_tap, size, combo; } orc_reg_t;
typedef struct { uint32_t op, a, b, c; } orc_step_t;
typedef struct { uint32_t kind, param; } orc_code_col_t;
typedef struct { uint32_t kind, a, b, c, e; } orc_data_col_t;
typedef struct { uint32_t first, a, b; } orc_acc_col_t;
typedef struct { uint32_t n_f; uint32_t col[3][4]; } orc_acc_fp_t; /* running product of up to three tuple fingerprints (blob section 8) */
/* the log-derivative argument (blob section 10): fractions numerator / sum of (challenge x linear form), four to an accumulator */
typedef struct { uint32_t coef, global, col; } orc_lf_term_t; /* canonical coefficient; public input + 1 or 0; column ref + 1 or 0 */
typedef struct { uint32_t n; const orc_lf_term_t* t; } orc_lf_t; /* points into the circuit's copy of the section */
typedef struct { uint32_t ch_kind, ch_idx; orc_lf_t lf; } orc_logup_part_t;
typedef struct { uint32_t table; orc_lf_t num; uint32_t n_parts; orc_logup_part_t parts[8]; } orc_logup_fraction_t;
typedef struct { uint32_t final_global; orc_logup_fraction_t fr[4]; } orc_logup_acc_t;

struct orc_circuit {
  uint32_t group_size[3];
  uint32_t n_taps; orc_tap_t* taps;
  uint32_t n_regs; orc_reg_t* regs;
  uint32_t n_combos; uint32_t* combo_begin; uint32_t* combo_backs; /* combo_begin has n_combos+1 entries */
  uint32_t group_tap_begin[4];
  uint32_t n_global, n_mix; uint32_t* global_cols; /* global k = DATA column global_cols[k] at row 0 */
  uint32_t n_steps, ret; orc_step_t* steps;
  uint32_t n_fp_vars, n_mix_vars;
  uint32_t n_code; orc_code_col_t* code_cols;
  uint32_t n_data; orc_data_col_t* data_cols;
  uint32_t n_acc; orc_acc_col_t* acc_cols;
  uint32_t n_acc_fp; orc_acc_fp_t* acc_fp;
  uint32_t n_late;                       /* the last n_late public inputs enter the transcript after the DATA commitment */
  uint32_t n_logup, n_chain, n_tables;   /* accumulators of the log-derivative argument; how many are links of the chain */
  orc_logup_acc_t* logup;
  uint32_t table_col[8], table_kind[8];  /* DATA column that holds a table's multiplicities; 1 = range-16, 2 = byte-AND */
  uint32_t* logup_words;                 /* the section itself (the linear forms point into it) */
  uint32_t period, n_periodic; uint32_t* periodic;  /* CODE columns of kind 6: n_periodic x period canonical values */
  uint32_t has_sponge, sponge_code, sponge_data, sponge_global;  /* the in-circuit Poseidon2 sponge: its first columns, its digest's inputs */
  uint8_t info[16]; /* circuit ProtocolInfo tag (risc0 `CIRCUIT_INFO`), 16 bytes */
};

/* poly_ext opcodes (risc0-zkp adapter.rs `PolyExtStep`) */
enum { OP_CONST = 0, OP_GET = 2, OP_GET_GLOBAL = 3, OP_ADD = 4, OP_SUB = 5, OP_MUL = 6, OP_TRUE = 7, OP_AND_EQZ = 8, OP_AND_COND = 9 };
#define REF_GROUP(r) ((r) >> 28)
#define REF_BACK(r) (((r) >> 20) & 0xffu)
#define REF_COL(r) ((r) & 0xfffffu)
#endif
