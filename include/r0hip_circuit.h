/* Circuit blob consumed by r0h_circuit_load / r0h_circuit_emit_hip.
 *
 * Everything circuit-specific in the reference's prover is machine-generated data compiled into
 * risc0-circuit-rv32im 4.0.4 (tap table `TAPSET`, constraint program `POLY_EXT`, step functions); here it crosses
 * the C ABI as one little-endian u32 stream so that a real circuit can be dropped in without touching the kernels.
 *
 *   word 0  magic 0x31433052 ("R0C1")      word 1  version (1)      word 2  number of sections
 *   section = [tag, n_words, payload...]
 *
 *   GROUPS  (1): size of tap groups ACCUM(0), CODE(1), DATA(2) in columns
 *   TAPS    (2): n_taps, then (group, offset, back) sorted ascending; a register = all taps of one (group, offset);
 *                every column must own a tap with back 0; combos (distinct back lists) are derived on load
 *   GLOBALS (3): n_global, n_mix, then (synthetic circuits only) n_global DATA column indices (global k = that column at row 0)
 *   POLY    (4): n_steps, ret (mix var), then (op, a, b, c):
 *                  0 CONST a=canonical value      2 GET a=tap index       3 GET_GLOBAL a=0 global | 1 mix, b=offset
 *                  4 ADD / 5 SUB / 6 MUL a,b=fp vars                      7 TRUE
 *                  8 AND_EQZ a=mix var, b=fp var                          9 AND_COND a=mix var, b=fp var, c=inner mix var
 *                fp vars and mix vars are numbered separately in creation order (risc0-zkp adapter.rs PolyExtStep)
 *   WITGEN  (5): [optional, with ACCUM: the synthetic column program; circuits imported from risc0 omit both] n_code, (kind, param) per CODE column: 0 first-row flag, 1 last-row flag, 2 row counter, 3 fixed random,
 *                4 the 16-bit range table (row r < 2^16: r, else 0), 5 the byte-AND table (row r = a + 256 b < 2^16: 2^24 + r + 65536 (a & b), else 2^24),
 *                6 a periodic schedule (param = a column of the PERIODIC section: row r holds values[param][r mod period] while a whole period
 *                still fits below the trace's end, else 0);
 *                n_data, (kind, a, b, c, e) per DATA column: 0 seeded random, 1 a*b+e, 2 a*b*c+e with refs
 *                ref = group<<28 | back<<20 | column (group 1 or 2; DATA refs point at lower-numbered columns)
 *   INFO    (7): optional, 4 words = the circuit's 16-byte ProtocolInfo tag committed into the transcript (risc0 `CIRCUIT_INFO`)
 *   ACCUM   (6): n_acc, (first_code_col, a_data_col, b_data_col): extension column j (ACCUM columns 4j..4j+3) is the
 *                running product of (mix[8j..8j+4) + a + mix[8j+4..8j+8) * b) from row 0
 *   ACCUM_FP (8): [instead of ACCUM: the memory-consistency argument of the trace circuit] n_acc, then 13 words per accumulator:
 *                n_f (1..3) and three (addr, lo, hi, t) quadruples of DATA columns; extension column j is the running product
 *                from row 0 of prod_{f < n_f} (alpha - addr_f - b1 lo_f - b2 hi_f - b3 t_f) with alpha = mix[0..4),
 *                b1 = mix[4..8), b2 = mix[8..12), b3 = mix[12..16) shared by all accumulators (n_mix = 16)
 *   LATE    (9): n_late: the last n_late public inputs are absorbed by the transcript AFTER the DATA group is committed (and before
 *                the accumulation mix is drawn) instead of at the start -- inputs that depend on commitments made outside this proof
 *                (the trace circuit's session-wide challenge and the segment's sum under it)
 *   LOGUP  (10): [instead of ACCUM: a log-derivative argument -- lookups, memory tuples, session tuples] n_acc, n_tables, then
 *                (DATA column, kind) per looked-up table (the column receives the multiplicities; kind 1 = range-16, 2 = byte-AND),
 *                then per accumulator: n_fractions (<= 4), final (0xffffffff: a link of the chain; else the index of the first of
 *                the four public inputs its total is), and per fraction: table (0 none / 1 / 2: a lookup into that table, whose
 *                value is minus the form of part 1 [minus 2^24 for kind 2]), the numerator (a linear form), n_parts, and per
 *                part (challenge kind 0 one / 1 mix element / 2 four public inputs, its index, a linear form): the denominator is
 *                the sum over parts of challenge x form.  A linear form is n_terms x (canonical coefficient, public input + 1 or
 *                0, column ref + 1 or 0 for the constant one), ref = group << 28 | column.  Extension column j (ACCUM columns
 *                4j..4j+3): chain links run ONE sum through the row's accumulators and on through the rows, wrapping around the
 *                end of the trace (so its total is zero); an accumulator with a public total runs alone and wraps with it
 *   PERIODIC (11): period, n_cols, then n_cols x period canonical values: what CODE columns of kind 6 repeat
 *   SPONGE (12): [the in-circuit Poseidon2 sponge of the recursion circuit: tools/sponge_component.py] first CODE column (28 columns: rc[24],
 *                sel_mix, sel_full, sel_part, sel_last), first DATA column (65 columns: st[24], aux[24], in[16], act), first of the 8 public
 *                inputs the sponge's digest is tied to.  The prover fills the DATA columns from the words it hashes (r0h_sponge_trace)
 */
#ifndef R0HIP_CIRCUIT_H
#define R0HIP_CIRCUIT_H
#define R0H_BLOB_MAGIC 0x31433052u
#define R0H_SEC_GROUPS 1
#define R0H_SEC_TAPS 2
#define R0H_SEC_GLOBALS 3
#define R0H_SEC_POLY 4
#define R0H_SEC_WITGEN 5
#define R0H_SEC_ACCUM 6
#define R0H_SEC_INFO 7
#define R0H_SEC_ACCUM_FP 8
#define R0H_SEC_LATE 9
#define R0H_SEC_LOGUP 10
#define R0H_SEC_PERIODIC 11
#define R0H_SEC_SPONGE 12
#define R0H_SPONGE_DATA_COLUMNS 65
#define R0H_SPONGE_CODE_COLUMNS 28
#define R0H_SPONGE_PERIOD 30
#define R0H_TABLE_R16 1
#define R0H_TABLE_AND 2
#define R0H_TAG_AND (1u << 24)
#define R0H_OP_CONST 0
#define R0H_OP_GET 2
#define R0H_OP_GET_GLOBAL 3
#define R0H_OP_ADD 4
#define R0H_OP_SUB 5
#define R0H_OP_MUL 6
#define R0H_OP_TRUE 7
#define R0H_OP_AND_EQZ 8
#define R0H_OP_AND_COND 9
#endif
