// `Prover::prove(env, elf)` in the shape hyperfridge calls it (host/src/main.rs:420-423): execute the guest, cut the run
// into segments, prove every segment, assemble the composite receipt.  risc0-zkvm 3.0.5 `LocalProver::prove` = executor
// (`ExecutorImpl::run` -> Session{segments}) + `prove_session` (one `prove_segment` per segment) + CompositeReceipt.
//
// What each stage is here: the executor is csrc/rv32im.hip (RV32IM by the specification; this library's ecall ABI and cycle model,
// see there); `prove_segment` is the device-resident sequencer (csrc/prover.hip) at the trace size the segment needs; the claim of
// every segment (system states from the run, exit code, the journal's output digest on the last one) is bound to its seal through
// the public inputs (csrc/claim.hip).  What it is NOT: the witness is the circuit blob's synthetic column program with the claim
// planted -- it is not derived from the preflight trace, because the rv32im step functions (risc0-circuit-rv32im-sys) cannot be
// reproduced here (hard part 2).  The seal therefore proves "a satisfying trace of the loaded circuit exists whose
// public inputs name this claim", not "this program ran": the row a9 gap, stated wherever this entry point is described.
#include <string.h>

#include <memory>
#include <vector>

#include "circuit.hpp"
#include "receipt_types.hpp"

using namespace r0h;

extern "C" {

const char* r0h_prove_elf(r0h_ctx* ctx, const r0h_circuit* c, const uint8_t* elf, size_t elf_len, const uint32_t* input_words, size_t n_input, uint32_t segment_po2,
                          uint64_t max_cycles, r0h_receipt** receipt_out, uint8_t image_id_out[32], uint64_t* cycles_out) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(ctx && c && elf && receipt_out && (input_words || !n_input), "r0h_prove_elf: NULL argument");
  R0H_REQUIRE(segment_po2 >= 9 && segment_po2 <= R0H_MAX_PO2, "r0h_prove_elf: segment_po2 %u outside [9, %u]", segment_po2, R0H_MAX_PO2);
  R0H_REQUIRE(c->has_column_program, "r0h_prove_elf: the circuit has no column program to plant a claim into");
  R0H_REQUIRE(c->n_global >= 8, "r0h_prove_elf: the circuit exposes %u public inputs, a claim needs 8", c->n_global);
  // 1. execute and segment (no trace kept: the synthetic witness does not read it)
  r0h_vm* vm = nullptr;
  R0H_TRY(r0h_vm_new(&vm));
  struct VmGuard { r0h_vm* v; ~VmGuard() { r0h_vm_free(v); } } guard{vm};
  R0H_TRY(r0h_vm_load_elf(vm, elf, elf_len));
  R0H_TRY(r0h_vm_set_input(vm, input_words, n_input));
  r0h_vm_limits lim;
  memset(&lim, 0, sizeof lim);
  lim.segment_po2 = segment_po2;
  lim.max_cycles = max_cycles;
  int exit_kind = 0;
  uint32_t exit_code = 0;
  R0H_TRY(r0h_vm_run(vm, &lim, &exit_kind, &exit_code));
  R0H_REQUIRE(exit_kind != R0H_VM_LIMIT, "r0h_prove_elf: the guest did not halt within %llu cycles (session limit)", (unsigned long long)max_cycles);
  R0H_REQUIRE(exit_code == 0, "r0h_prove_elf: the guest exited with code %u", exit_code);  // `prove` is an Err for a failed guest
  if (cycles_out) *cycles_out = r0h_vm_cycles(vm);
  const uint8_t* journal; size_t journal_len;
  R0H_TRY(r0h_vm_journal(vm, &journal, &journal_len));
  // 2. prove every segment for its claim
  r0h_receipt* rc = nullptr;
  R0H_TRY(r0h_receipt_new(R0H_RECEIPT_COMPOSITE, journal, journal_len, &rc));
  std::unique_ptr<r0h_receipt, const char* (*)(r0h_receipt*)> rc_guard(rc, r0h_receipt_free);
  const size_t n_seg = r0h_vm_n_segments(vm);
  std::vector<uint32_t> seal((size_t)1 << 20), global(c->n_global);
  for (size_t i = 0; i < n_seg; i++) {
    r0h_vm_segment info;
    r0h_receipt_claim claim;
    R0H_TRY(r0h_vm_segment_info(vm, i, &info));
    R0H_TRY(r0h_vm_segment_claim(vm, i, &claim));
    uint32_t po2 = 9;  // the smallest trace that holds the segment's cycles
    while (((uint64_t)1 << po2) < info.user_cycles + info.paging_cycles) po2++;
    uint8_t cd[32];
    claim_digest(claim, cd);
    std::fill(global.begin(), global.end(), 0u);
    claim_globals(cd, global.data());
    const size_t n = (size_t)1 << po2;
    r0h_buf *code = nullptr, *data = nullptr;
    R0H_TRY(buf_alloc_pooled(ctx, (size_t)c->group_size[R0H_GROUP_CODE] * n * 4, &code));
    const char* err = buf_alloc_pooled(ctx, (size_t)c->group_size[R0H_GROUP_DATA] * n * 4, &data);
    size_t words = 0;
    if (!err) err = r0h_witgen_public(ctx, c, po2, 0x5E55 + i, global.data(), code, data);
    if (!err) err = r0h_prove_segment(ctx, c, po2, code, data, global.data(), seal.data(), seal.size(), &words);
    r0h_buf_free(code);
    if (data) r0h_buf_free(data);
    if (err) return err;
    R0H_TRY(r0h_receipt_add_segment_claim(rc, seal.data(), words, (uint32_t)i, &claim, nullptr));
  }
  // 3. the image id the verifier is given: digest of the state the run started from
  if (image_id_out) {
    r0h_vm_segment first;
    R0H_TRY(r0h_vm_segment_info(vm, 0, &first));
    system_state_digest(first.pre, image_id_out);
  }
  *receipt_out = rc_guard.release();
  return nullptr;
  R0H_GUARD_END
}

}  // extern "C"
