// lift + join behind the C ABI (risc0-zkvm 3.0.5 `ProverServer::{lift, join}` over
// risc0-circuit-recursion 4.0.4, Cargo.lock:3050-3085): `lift` turns a segment receipt into a recursion-circuit proof, `join` folds
// two of them into one whose claim is the composition of theirs -- {pre: a.pre, post: b.post, exit_code: b.exit_code,
// input: a.input, output: b.output}, refused unless a ends in SystemSplit exactly where b starts -- and the folds form the binary tree
// whose levels are the one inter-GPU exchange of the whole path (hyperfridge-r0_amd/recursion.py moves the nodes; nothing else).
//
// What a node's proof is, stated plainly: one STARK over the recursion circuit of this repository (circuits/recursion.r0c) made with the
// same kernels as a segment proof, whose 16 public inputs are (a) the 8 words naming the node's composed ReceiptClaim
// (r0h_claim_globals) and (b) the Poseidon2 digest of what it consumed (the segment seal, or the two child seals' digests).  (b) is
// computed INSIDE the proof: the circuit's sponge component (tools/sponge_component.py, blob section SPONGE) hashes the consumed words,
// held in witness cells, one Poseidon2 round per row, and ties the result to those public inputs -- a node whose witness holds other
// words than its digest names has no satisfying trace.  That is the first in-circuit step and the only one: the children's SEALS are
// still verified BESIDE the proof -- host threads run the verifier while the device proves -- because risc0's recursion circuit (whose
// programs are downloaded at build time upstream) cannot be reproduced here; a root node is a checkable tree of seals with a composed
// claim, not a succinct receipt.  r0h_node_verify checks one node's seal, control root and naming words.
#include <string.h>

#include <memory>
#include <thread>
#include <vector>

#include "circuit.hpp"
#include "receipt_types.hpp"

using namespace r0h;

struct r0h_node {
  std::vector<uint32_t> seal;
  r0h_receipt_claim claim;
};

struct r0h_recursor {
  r0h_ctx* ctx = nullptr;
  r0h_circuit* circuit = nullptr;          // the recursion-shaped circuit, loaded on ctx
  r0h_code_commit* code = nullptr;         // its CODE group at `po2`, committed once
  uint32_t po2 = 0;
  uint32_t root[8] = {0};                  // control root of the recursion circuit at po2
  std::vector<uint32_t> recursion_blob, segment_blob;
  std::vector<uint32_t> segment_roots;     // n records of 9 words [po2, root[8]]: control roots of the segment circuit
};

namespace {
bool same_state(const r0h_system_state& a, const r0h_system_state& b) {
  uint8_t da[32], db[32];
  system_state_digest(a, da);
  system_state_digest(b, db);
  return !memcmp(da, db, 32);
}

void naming_words(const r0h_receipt_claim& claim, uint32_t out[8]) {
  uint8_t cd[32];
  claim_digest(claim, cd);
  claim_globals(cd, out);
}

// one recursion-circuit proof with the given 16 public inputs, on the recursor's context, while `checks` run on host threads
const char* prove_node(r0h_recursor* rc, const uint32_t publics[16], const uint32_t* consumed, size_t n_consumed, std::vector<uint32_t>& seal) {
  r0h_ctx* ctx = rc->ctx;
  const size_t n = (size_t)1 << rc->po2;
  r0h_buf *code = nullptr, *data = nullptr;
  R0H_TRY(buf_alloc_pooled(ctx, (size_t)rc->circuit->group_size[R0H_GROUP_CODE] * n * 4, &code));
  const char* err = buf_alloc_pooled(ctx, (size_t)rc->circuit->group_size[R0H_GROUP_DATA] * n * 4, &data);
  // the rest of the witness is the circuit's synthetic column program: any deterministic seed
  const uint64_t seed = (uint64_t)publics[0] | (uint64_t)publics[8] << 32;
  if (!err) err = r0h_witgen_public(ctx, rc->circuit, rc->po2, seed, publics, code, data);
  // the sponge rows over what this node consumes: the digest the circuit computes from them is publics[8..16)
  if (!err) err = sponge_plant(ctx, rc->circuit, rc->po2, consumed, n_consumed, data);
  seal.resize((size_t)1 << 19);
  size_t words = 0;
  if (!err) err = r0h_prove_segment_committed(ctx, rc->circuit, rc->po2, rc->code, data, publics, seal.data(), seal.size(), &words);
  r0h_buf_free(code);
  if (data) r0h_buf_free(data);
  if (err) return err;
  seal.resize(words);
  return nullptr;
}

struct Check {  // a seal this step consumes, verified on a host thread while the device proves
  const uint32_t* blob; size_t blob_words; const uint32_t* seal; size_t seal_words; const uint32_t* root; const char* what;
  int verdict = -1;
  const char* err = nullptr;
  void run() { err = r0h_verify_seal_bound(blob, blob_words, nullptr, nullptr, seal, seal_words, root, &verdict, nullptr, nullptr); }
};

const char* prove_checked(r0h_recursor* rc, std::vector<Check>& checks, const uint32_t publics[16], const uint32_t* consumed, size_t n_consumed,
                          std::vector<uint32_t>& seal) {
  std::vector<std::thread> threads;
  for (Check& c : checks) threads.emplace_back([&c] { c.run(); });
  const char* err = prove_node(rc, publics, consumed, n_consumed, seal);
  for (std::thread& t : threads) t.join();
  for (Check& c : checks) {
    if (c.err && !err) err = c.err;
    else if (c.err) r0h_free_error(c.err);
  }
  if (err) return err;
  for (Check& c : checks) R0H_REQUIRE(c.verdict == R0H_VERIFY_OK, "%s: the seal to be consumed does not verify: %s", c.what, r0h_verify_reason(c.verdict));
  return nullptr;
}
}  // namespace

extern "C" {

const char* r0h_recursor_new(r0h_ctx* ctx, const uint32_t* recursion_blob, size_t recursion_words, const char* code_object_path, uint32_t po2,
                             const uint32_t* segment_blob, size_t segment_words, const uint32_t* segment_control_roots, size_t n_roots, r0h_recursor** out) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(ctx && recursion_blob && segment_blob && out && (segment_control_roots || !n_roots), "r0h_recursor_new: NULL argument");
  R0H_REQUIRE(po2 >= 9 && po2 <= R0H_MAX_PO2, "r0h_recursor_new: po2 %u outside [9, %u]", po2, R0H_MAX_PO2);
  std::unique_ptr<r0h_recursor, const char* (*)(r0h_recursor*)> rc(new r0h_recursor(), r0h_recursor_free);
  rc->ctx = ctx; rc->po2 = po2;
  ctx_retain(ctx);
  rc->recursion_blob.assign(recursion_blob, recursion_blob + recursion_words);
  rc->segment_blob.assign(segment_blob, segment_blob + segment_words);
  rc->segment_roots.assign(segment_control_roots, segment_control_roots + 9 * n_roots);
  {
    r0h_circuit seg;  // the leaves' circuit is only ever handed to the verifier: parse it once to refuse a malformed blob early
    R0H_TRY(parse_blob(&seg, segment_blob, segment_words));
    R0H_REQUIRE(seg.n_global >= 8, "r0h_recursor_new: the segment circuit exposes %u public inputs, a claim needs 8", seg.n_global);
  }
  R0H_TRY(r0h_circuit_load(ctx, recursion_blob, recursion_words, code_object_path, &rc->circuit));
  R0H_REQUIRE(rc->circuit->n_global == 16 && rc->circuit->has_column_program, "r0h_recursor_new: a recursion circuit exposes 16 public inputs (claim words, digest of what it consumed) and carries a column program; this one has %u",
              rc->circuit->n_global);
  R0H_REQUIRE(rc->circuit->has_sponge && rc->circuit->sponge_global == 8, "r0h_recursor_new: a recursion circuit computes the digest of what a node consumed in-circuit (blob section SPONGE, public inputs 8..15); this one does not");
  const size_t n = (size_t)1 << po2;
  r0h_buf *code = nullptr, *data = nullptr;
  R0H_TRY(buf_alloc_pooled(ctx, (size_t)rc->circuit->group_size[R0H_GROUP_CODE] * n * 4, &code));
  const char* err = buf_alloc_pooled(ctx, (size_t)rc->circuit->group_size[R0H_GROUP_DATA] * n * 4, &data);
  if (!err) err = r0h_witgen(ctx, rc->circuit, po2, 0, code, data, nullptr);
  if (!err) err = r0h_code_commit_new(ctx, code, rc->circuit->group_size[R0H_GROUP_CODE], po2, &rc->code);
  r0h_buf_free(code);
  if (data) r0h_buf_free(data);
  if (err) return err;
  R0H_TRY(r0h_code_commit_root(rc->code, rc->root));
  *out = rc.release();
  return nullptr;
  R0H_GUARD_END
}

const char* r0h_recursor_free(r0h_recursor* rc) {
  if (!rc) return nullptr;
  if (rc->code) r0h_code_commit_free(rc->code);
  if (rc->circuit) r0h_circuit_free(rc->circuit);
  r0h_ctx* ctx = rc->ctx;
  delete rc;
  if (ctx) ctx_release(ctx);
  return nullptr;
}

const char* r0h_recursor_control_root(const r0h_recursor* rc, uint32_t root_out[8]) {
  R0H_REQUIRE(rc && root_out, "r0h_recursor_control_root: NULL argument");
  memcpy(root_out, rc->root, 32);
  return nullptr;
}

const char* r0h_lift(r0h_recursor* rc, const uint32_t* seal, size_t seal_words, const r0h_receipt_claim* claim, r0h_node** out) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(rc && seal && claim && out, "r0h_lift: NULL argument");
  r0h_circuit seg;
  R0H_TRY(parse_blob(&seg, rc->segment_blob.data(), rc->segment_blob.size()));
  R0H_REQUIRE(seal_words > (size_t)seg.n_global && seal[seg.n_global] < P, "r0h_lift: the segment seal is truncated");
  // `SegmentReceipt::verify_integrity`: the seal names the claim it is lifted for ...
  uint32_t publics[16];
  naming_words(*claim, publics);
  R0H_REQUIRE(!memcmp(publics, seal, 32), "r0h_lift: the segment seal's public inputs do not name this claim");
  // a trace-circuit seal also says where its run starts and stops and how it ends: the claim it is lifted under must say the same
  // (r0h_receipt_verify refuses a receipt otherwise; a root built from lifted nodes must not carry a claim that check would refuse)
  if (is_trace_circuit(seg)) R0H_REQUIRE(trace_seal_carries_claim(seal, *claim), "r0h_lift: the segment seal's first / last pc, way of ending or exit code are not this claim's");
  // ... and verifies against the control root of its trace size, when the recursor was given one (else against the circuit alone)
  const uint32_t po2 = dec(seal[seg.n_global]);
  const uint32_t* root = nullptr;
  for (size_t k = 0; k < rc->segment_roots.size() / 9; k++)
    if (rc->segment_roots[9 * k] == po2) root = rc->segment_roots.data() + 9 * k + 1;
  R0H_REQUIRE(root || rc->segment_roots.empty(), "r0h_lift: no control root known for a segment of 2^%u rows", po2);
  R0H_TRY(r0h_seal_digest(seal, seal_words, publics + 8));
  std::vector<Check> checks(1);
  checks[0] = Check{rc->segment_blob.data(), rc->segment_blob.size(), seal, seal_words, root, "lift"};
  std::unique_ptr<r0h_node> node(new r0h_node());
  node->claim = *claim;
  R0H_TRY(prove_checked(rc, checks, publics, seal, seal_words, node->seal));
  *out = node.release();
  return nullptr;
  R0H_GUARD_END
}

const char* r0h_join(r0h_recursor* rc, const r0h_node* a, const r0h_node* b, r0h_node** out) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(rc && a && b && out, "r0h_join: NULL argument");
  // risc0 `ReceiptClaim::join`: a must stop in a system split exactly where b starts.  One more case here: b is a node of closing rows
  // only (the trace circuit's session may end in segments without cycles: pre == post, SystemSplit, no output) -- it follows whatever
  // a ends in, the run's last segment included, and the composed claim ends the way a does
  static const uint8_t no_output[32] = {0};
  const bool b_idle = same_state(b->claim.pre, b->claim.post) && b->claim.exit_system == 2 && b->claim.exit_user == 0 && !memcmp(b->claim.output_digest, no_output, 32);
  R0H_REQUIRE(b_idle || (a->claim.exit_system == 2 && a->claim.exit_user == 0), "r0h_join: the left node does not end in SystemSplit: nothing can follow it");
  R0H_REQUIRE(same_state(a->claim.post, b->claim.pre), "r0h_join: the left node's post-state is not the right node's pre-state: these two do not follow one another");
  uint32_t names[16];
  for (int side = 0; side < 2; side++) {  // each child's seal names the claim it is carried with
    const r0h_node* nd = side ? b : a;
    R0H_REQUIRE(nd->seal.size() > 17, "r0h_join: a child seal is truncated");
    naming_words(nd->claim, names);
    R0H_REQUIRE(!memcmp(names, nd->seal.data(), 32), "r0h_join: the %s node's seal does not name the claim it is carried with", side ? "right" : "left");
  }
  std::unique_ptr<r0h_node> node(new r0h_node());
  node->claim = a->claim;             // pre, input from the left ...
  node->claim.post = b->claim.post;   // ... post, exit code, output from the right (a node of closing rows only leaves the left's)
  if (!(b_idle && a->claim.exit_system != 2)) {
    node->claim.exit_system = b->claim.exit_system;
    node->claim.exit_user = b->claim.exit_user;
    memcpy(node->claim.output_digest, b->claim.output_digest, 32);
  }
  uint32_t publics[16], children[16];
  naming_words(node->claim, publics);
  R0H_TRY(r0h_seal_digest(a->seal.data(), a->seal.size(), children));
  R0H_TRY(r0h_seal_digest(b->seal.data(), b->seal.size(), children + 8));
  R0H_TRY(r0h_seal_digest(children, 16, publics + 8));
  std::vector<Check> checks(2);
  checks[0] = Check{rc->recursion_blob.data(), rc->recursion_blob.size(), a->seal.data(), a->seal.size(), rc->root, "join (left)"};
  checks[1] = Check{rc->recursion_blob.data(), rc->recursion_blob.size(), b->seal.data(), b->seal.size(), rc->root, "join (right)"};
  R0H_TRY(prove_checked(rc, checks, publics, children, 16, node->seal));
  *out = node.release();
  return nullptr;
  R0H_GUARD_END
}

// a node as it arrives from another rank: the seal and the claim it is carried with (checked when it is joined or verified)
const char* r0h_node_new(const uint32_t* seal, size_t seal_words, const r0h_receipt_claim* claim, r0h_node** out) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(seal && claim && out, "r0h_node_new: NULL argument");
  std::unique_ptr<r0h_node> node(new r0h_node());
  node->seal.assign(seal, seal + seal_words);
  node->claim = *claim;
  *out = node.release();
  return nullptr;
  R0H_GUARD_END
}
const char* r0h_node_free(r0h_node* node) {
  delete node;
  return nullptr;
}
const char* r0h_node_seal(const r0h_node* node, const uint32_t** seal, size_t* seal_words) {
  R0H_REQUIRE(node && seal && seal_words, "r0h_node_seal: NULL argument");
  *seal = node->seal.data();
  *seal_words = node->seal.size();
  return nullptr;
}
const char* r0h_node_claim(const r0h_node* node, r0h_receipt_claim* claim_out) {
  R0H_REQUIRE(node && claim_out, "r0h_node_claim: NULL argument");
  *claim_out = node->claim;
  return nullptr;
}
// One node on its own: its seal verifies against the recursion circuit bound to `control_root` (NULL: unbound), and its first eight
// public inputs name the claim it is carried with.  *ok_out = 1 / 0; pure host code.  (What it consumed is vouched for by whoever
// made it: see the header of this file.)
const char* r0h_node_verify(const uint32_t* recursion_blob, size_t blob_words, const uint32_t* control_root, const r0h_node* node, int* ok_out) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(recursion_blob && node && ok_out, "r0h_node_verify: NULL argument");
  *ok_out = 0;
  int verdict = -1;
  R0H_TRY(r0h_verify_seal_bound(recursion_blob, blob_words, nullptr, nullptr, node->seal.data(), node->seal.size(), control_root, &verdict, nullptr, nullptr));
  if (verdict != R0H_VERIFY_OK || node->seal.size() < 16) return nullptr;
  uint32_t names[8];
  naming_words(node->claim, names);
  *ok_out = !memcmp(names, node->seal.data(), 32);
  return nullptr;
  R0H_GUARD_END
}

}  // extern "C"
