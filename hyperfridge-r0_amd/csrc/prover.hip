// The segment-proof sequencer: owns the Fiat-Shamir transcript on the host and drives the device operations.
// Replaces risc0-circuit-rv32im 4.0.4 prove/hal/mod.rs (`SegmentProver::prove`: seed transcript, commit CODE and DATA,
// draw the accumulation mix, accumulate, finalize) and risc0-zkp 3.0.4 prove/{prover.rs, poly_group.rs, merkle.rs,
// fri.rs, write_iop.rs} + core/hash/poseidon2/rng.rs -- SURVEY.md 3.4 steps 3-13, 8(a) a8, a15-a17.
// This is what host/src/main.rs:423 (`prover.prove(env, HYPERFRIDGE_ELF)`) spends its time in, once per segment.
//
// Device-resident throughout: witness, coefficients, evaluations, Merkle nodes, combos.  What crosses to the host per
// segment: Merkle tops + roots (<= 2 KB each), ~500 tap evaluations, the FRI final polynomial (4 KB) and the 50
// query openings (gathered on the device into one packed buffer per tree).  Upstream pulls the combos back to the
// host for the DEEP division; here the division is a device scan (r0h_poly_divide).
#include <memory>

#include "circuit.hpp"

namespace r0h {

// flat view of the circuit tables the sequencer walks
struct CircuitView {
  uint32_t group_size[3];
  uint32_t n_taps, n_regs, n_combos, n_global, n_mix;
  std::vector<uint32_t> tap_offset, tap_back, reg_group, reg_offset, reg_first, reg_size, reg_combo;
  const uint32_t* combo_begin; const uint32_t* combo_backs;
  uint32_t group_tap_begin[4];
  const uint32_t* blob; size_t blob_words;
};
static void circuit_view(const r0h_circuit* c, CircuitView* v) {
  memcpy(v->group_size, c->group_size, sizeof v->group_size);
  v->n_taps = (uint32_t)c->taps.size(); v->n_regs = (uint32_t)c->regs.size(); v->n_combos = (uint32_t)c->combo_begin.size() - 1;
  v->n_global = c->n_global; v->n_mix = c->n_mix;
  for (const Tap& t : c->taps) { v->tap_offset.push_back(t.offset); v->tap_back.push_back(t.back); }
  for (const Reg& r : c->regs) {
    v->reg_group.push_back(r.group); v->reg_offset.push_back(r.offset); v->reg_first.push_back(r.first_tap);
    v->reg_size.push_back(r.size); v->reg_combo.push_back(r.combo);
  }
  v->combo_begin = c->combo_begin.data(); v->combo_backs = c->combo_backs.data();
  memcpy(v->group_tap_begin, c->group_tap_begin, sizeof v->group_tap_begin);
  v->blob = c->blob.data(); v->blob_words = c->blob.size();
}

static unsigned log2u(size_t x) { unsigned n = 0; while (((size_t)1 << n) < x) n++; return n; }

// ------------------------------------------------------------------ transcript
struct Rng {
  const P2Consts* k;
  uint32_t cells[P2_CELLS];
  uint32_t pool_used;
  explicit Rng(const P2Consts* kk) : k(kk), pool_used(0) { memset(cells, 0, sizeof cells); }
  void mix(const uint32_t digest[8]) {
    if (pool_used != 0) { p2_mix_host(*k, cells); pool_used = 0; }
    for (int i = 0; i < 8; i++) cells[i] = add(cells[i], digest[i]);  // digests come from the device or the host sponge: canonical
    p2_mix_host(*k, cells);
  }
  uint32_t elem() {
    if (pool_used == P2_RATE) { p2_mix_host(*k, cells); pool_used = 0; }
    return cells[pool_used++];
  }
  Fp4 ext() { Fp4 r; for (int i = 0; i < 4; i++) r.e[i] = elem(); return r; }
  uint32_t bits(uint32_t n) {
    uint32_t val = dec(elem());
    for (int i = 0; i < 3; i++) { uint32_t nv = dec(elem()); if (val == 0) val = nv; }
    return val & (uint32_t)(((uint64_t)1 << n) - 1);
  }
};
struct WriteIop {
  std::vector<uint32_t> proof;
  Rng rng;
  explicit WriteIop(const P2Consts* k) : rng(k) {}
  void write(const uint32_t* w, size_t n) { proof.insert(proof.end(), w, w + n); }
  void commit(const uint32_t digest[8]) { rng.mix(digest); }
};

// ------------------------------------------------------------------ Merkle
struct MerkleParams {
  size_t row_size, col_size, layers, top_layer, top_size;
  MerkleParams(size_t rows, size_t cols) : row_size(rows), col_size(cols) {
    layers = log2u(rows);
    top_layer = 0;
    for (size_t i = 1; i < layers; i++) {
      if (((size_t)1 << i) > R0H_QUERIES) break;
      top_layer = i;
    }
    top_size = (size_t)1 << top_layer;
  }
  size_t path_digests() const { return layers - top_layer; }
  size_t opening_words() const { return col_size + 8 * path_digests(); }
};

// out[q] = column values at row idx[q] followed by the sibling digests up to (excluding) the top layer
__global__ void merkle_open_kernel(uint32_t* __restrict__ out, const uint32_t* __restrict__ matrix, const uint32_t* __restrict__ nodes,
                                   const uint32_t* __restrict__ idx, uint32_t row_size, uint32_t col_size, uint32_t n_path) {
  const uint32_t q = blockIdx.x, row = idx[q];
  uint32_t* dst = out + (size_t)q * (col_size + 8 * n_path);
  for (uint32_t i = threadIdx.x; i < col_size; i += blockDim.x) dst[i] = matrix[(size_t)i * row_size + row];
  for (uint32_t w = threadIdx.x; w < 8 * n_path; w += blockDim.x) {
    uint32_t level = w >> 3, node = ((row + row_size) >> level) ^ 1u;
    dst[col_size + w] = nodes[(size_t)node * 8 + (w & 7)];
  }
}

struct Tree {
  MerkleParams mp;
  r0h_buf* nodes = nullptr;
  const r0h_buf* matrix = nullptr;
  Tree(size_t rows, size_t cols) : mp(rows, cols) {}
};

struct Scope {  // frees device buffers on every exit path
  std::vector<r0h_buf*> bufs;
  ~Scope() { for (r0h_buf* b : bufs) r0h_buf_free(b); }
  const char* alloc(r0h_ctx* ctx, size_t bytes, r0h_buf** out) {
    R0H_TRY(buf_alloc_pooled(ctx, bytes, out));
    bufs.push_back(*out);
    return nullptr;
  }
  void release(r0h_buf* b) {
    for (size_t i = 0; i < bufs.size(); i++)
      if (bufs[i] == b) { bufs.erase(bufs.begin() + i); r0h_buf_free(b); return; }
  }
};

static void phase(r0h_ctx* ctx, const char* name) {
  Profile& p = ctx->prof;
  size_t i = p.names.size();
  if (p.events.size() <= i) {
    hipEvent_t e;
    hipEventCreate(&e);
    p.events.push_back(e);
  }
  hipEventRecord(p.events[i], ctx->stream);
  p.names.push_back(name);
}

static const char* tree_build(r0h_ctx* ctx, Scope& sc, Tree& t, const r0h_buf* matrix) {
  t.matrix = matrix;
  R0H_TRY(sc.alloc(ctx, t.mp.row_size * 2 * 32, &t.nodes));
  return r0h_merkle_build(ctx, t.nodes, matrix, (uint32_t)t.mp.row_size, (uint32_t)t.mp.col_size);
}
static const char* tree_commit(r0h_ctx* ctx, Tree& t, WriteIop& io) {
  std::vector<uint32_t> host(2 * t.mp.top_size * 8);
  R0H_TRY(r0h_buf_d2h(ctx, t.nodes, 32, host.data() + 8, (2 * t.mp.top_size - 1) * 32));
  io.write(host.data() + t.mp.top_size * 8, t.mp.top_size * 8);
  io.commit(host.data() + 8);
  return nullptr;
}
// open `n_q` rows: returns packed openings on the host
static const char* tree_open(r0h_ctx* ctx, Scope& sc, const Tree& t, const r0h_buf* d_idx, uint32_t n_q, std::vector<uint32_t>& host) {
  r0h_buf* packed = nullptr;
  const size_t words = t.mp.opening_words();
  R0H_TRY(sc.alloc(ctx, (size_t)n_q * words * 4, &packed));
  hipLaunchKernelGGL(merkle_open_kernel, dim3(n_q), dim3(256), 0, ctx->stream, u32(packed), u32(t.matrix), u32(t.nodes), u32(d_idx),
                     (uint32_t)t.mp.row_size, (uint32_t)t.mp.col_size, (uint32_t)t.mp.path_digests());
  hipError_t e = hipGetLastError();
  R0H_REQUIRE(e == hipSuccess, "merkle_open_kernel: %s", hipGetErrorString(e));
  host.resize((size_t)n_q * words);
  R0H_TRY(r0h_buf_d2h(ctx, packed, 0, host.data(), host.size() * 4));
  sc.release(packed);
  return nullptr;
}

// ------------------------------------------------------------------ poly groups
struct Group {
  r0h_buf* coeffs = nullptr;     // bit-reversed, zk-shifted coefficients
  r0h_buf* evaluated = nullptr;  // [count][4N]
  uint32_t count = 0;
  Tree tree;
  Group(uint32_t cnt, size_t domain) : count(cnt), tree(domain, cnt) {}
};
// coeffs hold bit-reversed, zk-shifted coefficients: evaluate on 4N, commit, flip coeffs to natural order
static const char* group_finish(r0h_ctx* ctx, Scope& sc, Group& g, uint32_t po2) {
  R0H_TRY(sc.alloc(ctx, ((size_t)g.count << (po2 + 2)) * 4, &g.evaluated));
  R0H_TRY(r0h_batch_expand_into_evaluate_ntt(ctx, g.evaluated, g.coeffs, g.count, po2, 2));
  // upstream flips the coefficients to natural order here; the sequencer keeps them bit-reversed instead (evaluate-at-z
  // uses permuted power tables, the FRI mix is order-agnostic) and flips only the handful of mixed combos
  return tree_build(ctx, sc, g.tree, g.evaluated);
}
static const char* group_from_witness(r0h_ctx* ctx, Scope& sc, Group& g, const r0h_buf* witness, uint32_t po2) {
  const size_t bytes = ((size_t)g.count << po2) * 4;
  R0H_REQUIRE(bytes <= witness->bytes, "prove_segment: witness buffer holds fewer than %u columns of 2^%u", g.count, po2);
  R0H_TRY(sc.alloc(ctx, bytes, &g.coeffs));
  // out-of-place iNTT straight from the witness (no staging copy), zk shift fused into its last pass
  R0H_TRY(interpolate_ntt(ctx, g.coeffs, witness, g.count, po2, true));
  return group_finish(ctx, sc, g, po2);
}

// Lagrange interpolation of a handful of extension points (host; a register has at most a few taps)
static void poly_interpolate(Fp4* out, const Fp4* xs, const Fp4* ys, uint32_t n) {
  std::vector<Fp4> acc(n, fp4_zero()), basis(n + 1);
  for (uint32_t i = 0; i < n; i++) {
    uint32_t deg = 0;
    basis[0] = fp4_one();
    Fp4 denom = fp4_one();
    for (uint32_t j = 0; j < n; j++) {
      if (j == i) continue;
      basis[deg + 1] = fp4_zero();
      for (uint32_t k = deg + 1; k-- > 0;) {
        basis[k + 1] = basis[k + 1] + basis[k];
        basis[k] = fp4_zero() - basis[k] * xs[j];
      }
      deg++;
      denom = denom * (xs[i] - xs[j]);
    }
    Fp4 s = ys[i] * fp4_inv(denom);
    for (uint32_t k = 0; k < n; k++) acc[k] = acc[k] + basis[k] * s;
  }
  for (uint32_t k = 0; k < n; k++) out[k] = acc[k];
}

// combos[row 0..size) of one combo -= cur * coeff_u (tiny host-driven fix-up of the first few coefficients)
__global__ void sub_head_kernel(uint32_t* __restrict__ combos, const uint32_t* __restrict__ fix /* (word index, value) pairs */, uint32_t n) {
  uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) combos[fix[2 * k]] = sub(combos[fix[2 * k]], fix[2 * k + 1]);
}

}  // namespace r0h

// The committed CODE group of a program at one trace size: bit-reversed zk-shifted coefficients, evaluations on the 4N coset, Merkle
// nodes, and the host copies of what a commit sends to the transcript (top layer, root).  CODE depends on (circuit, po2) only --
// risc0 keeps one control root per po2 for exactly that reason -- so a prover commits it once and every segment of that size
// reads it: three device buffers that no kernel of a proof writes.  Complete (stream-synchronised) when r0h_code_commit returns,
// hence readable from any context of the same device.
struct r0h_code_commit {
  r0h_ctx* ctx = nullptr;
  uint32_t count = 0, po2 = 0;
  r0h_buf *coeffs = nullptr, *evaluated = nullptr, *nodes = nullptr;
  r0h_buf* witness = nullptr;  // the CODE columns themselves (a log-derivative accumulation reads its tables from them); may be absent
  std::vector<uint32_t> top;  // the top layer as tree_commit writes it to the seal
  uint32_t root[8] = {0};
};

// A proof in flight between r0h_proof_begin (CODE and DATA committed, accumulation mix drawn) and r0h_proof_finish.
// Mirrors risc0-zkp's `Prover` object between `commit_group(DATA)` and `finalize`.
struct r0h_proof {
  r0h_ctx* ctx;
  const r0h_circuit* circ;
  uint32_t po2;
  r0h::CircuitView cv;
  r0h::Scope sc;
  r0h::WriteIop io;
  r0h::Group g_accum, g_code, g_data, g_check;
  std::vector<uint32_t> mix, global;
  size_t seal_globals_at = 0;   // where the seal's opening block of public inputs starts
  uint32_t data_root[8] = {0};  // the DATA group's Merkle root (what a session's common challenge is derived from)
  bool mix_drawn = false;
  r0h_proof(r0h_ctx* c, const r0h_circuit* ci, uint32_t p, const r0h::CircuitView& v)
      : ctx(c), circ(ci), po2(p), cv(v), io(&c->p2_host), g_accum(v.group_size[R0H_GROUP_ACCUM], (size_t)4 << p),
        g_code(v.group_size[R0H_GROUP_CODE], (size_t)4 << p), g_data(v.group_size[R0H_GROUP_DATA], (size_t)4 << p),
        g_check(R0H_CHECK_SIZE, (size_t)4 << p) {}
};

namespace r0h {

static const char* proof_late(r0h_proof& st, const uint32_t* late);
static const char* proof_begin(r0h_proof& st, const r0h_buf* code, const r0h_code_commit* cc, const r0h_buf* data, const uint32_t* global) {
  r0h_ctx* ctx = st.ctx;
  const r0h_circuit* circ = st.circ;
  const uint32_t po2 = st.po2;
  CircuitView& cv = st.cv;
  Scope& sc = st.sc;
  WriteIop& io = st.io;
  Group &g_code = st.g_code, &g_data = st.g_data;
  ctx->prof.names.clear();
  st.global.assign(global, global + cv.n_global);

  phase(ctx, "transcript_seed");
  {
    // risc0-circuit-rv32im prove/hal: the hashes of two 16-byte ProtocolInfo tags (one field element per byte) open the
    // transcript: the proof system's and the circuit's
    static const char proof_system_info[] = "RISC0_STARK:v1__";
    uint32_t e[16], d[8];
    for (int i = 0; i < 16; i++) e[i] = enc((uint8_t)proof_system_info[i]);
    p2_hash_elems_host(ctx->p2_host, e, 16, d);
    io.commit(d);
    for (int i = 0; i < 16; i++) e[i] = enc(circ->info[i]);
    p2_hash_elems_host(ctx->p2_host, e, 16, d);
    io.commit(d);
    // the seal opens with every public input and po2; the transcript takes the early ones here and the late ones (inputs that depend
    // on commitments made outside this proof, R0H_SEC_LATE) after the DATA group is committed
    const uint32_t n_early = cv.n_global - circ->n_late;
    std::vector<uint32_t> gv(global, global + n_early);
    for (uint32_t w : gv) R0H_REQUIRE(w < P, "prove_segment: global word not canonical");
    gv.push_back(enc(po2));
    p2_hash_elems_host(ctx->p2_host, gv.data(), gv.size(), d);
    io.commit(d);
    st.seal_globals_at = io.proof.size();
    io.write(global, cv.n_global);
    const uint32_t po2_word = enc(po2);
    io.write(&po2_word, 1);
  }

  phase(ctx, "commit_code");
  if (cc) {  // committed ahead of time: the group's buffers are the cache's (not this proof's to free), the transcript sees the same words
    R0H_REQUIRE(cc->po2 == po2 && cc->count == g_code.count, "prove_segment: the CODE commitment is for %u columns of 2^%u rows, this proof needs %u of 2^%u",
                cc->count, cc->po2, g_code.count, po2);
    R0H_REQUIRE(cc->ctx->device == ctx->device, "prove_segment: the CODE commitment lives on device %d, this context on device %d", cc->ctx->device, ctx->device);
    g_code.coeffs = cc->coeffs;
    g_code.evaluated = cc->evaluated;
    g_code.tree.nodes = cc->nodes;
    g_code.tree.matrix = cc->evaluated;
    io.write(cc->top.data(), cc->top.size());
    io.commit(cc->root);
  } else {
    R0H_TRY(group_from_witness(ctx, sc, g_code, code, po2));
    R0H_TRY(tree_commit(ctx, g_code.tree, io));
  }
  phase(ctx, "commit_data");
  R0H_TRY(group_from_witness(ctx, sc, g_data, data, po2));
  R0H_TRY(tree_commit(ctx, g_data.tree, io));
  R0H_TRY(r0h_buf_d2h(ctx, g_data.tree.nodes, 32, st.data_root, 32));
  if (!circ->n_late) return proof_late(st, nullptr);
  return nullptr;
}

// the late public inputs enter the transcript (and the seal's opening block), then the accumulation mix is drawn
static const char* proof_late(r0h_proof& st, const uint32_t* late) {
  const uint32_t n_late = st.circ->n_late, n_early = st.cv.n_global - n_late;
  R0H_REQUIRE(!st.mix_drawn, "r0h_proof_late: the accumulation mix has been drawn already");
  if (n_late) {
    R0H_REQUIRE(late, "r0h_proof_late: NULL argument");
    for (uint32_t i = 0; i < n_late; i++) R0H_REQUIRE(late[i] < P, "r0h_proof_late: word %u not canonical", i);
    memcpy(st.global.data() + n_early, late, (size_t)n_late * 4);
    memcpy(st.io.proof.data() + st.seal_globals_at + n_early, late, (size_t)n_late * 4);
    uint32_t d[8];
    p2_hash_elems_host(st.ctx->p2_host, late, n_late, d);
    st.io.commit(d);
  }
  st.mix.resize(st.cv.n_mix);
  for (uint32_t i = 0; i < st.cv.n_mix; i++) st.mix[i] = st.io.rng.elem();
  st.mix_drawn = true;
  phase(st.ctx, "accum");
  return nullptr;
}

static const char* proof_finish(r0h_proof& st, const r0h_buf* accum, std::vector<uint32_t>& seal) {
  R0H_REQUIRE(st.mix_drawn, "r0h_proof_finish: this circuit has late public inputs: r0h_proof_late comes first");
  r0h_ctx* ctx = st.ctx;
  const r0h_circuit* circ = st.circ;
  const uint32_t po2 = st.po2;
  const size_t n = (size_t)1 << po2, domain = n * R0H_INV_RATE;
  CircuitView& cv = st.cv;
  Scope& sc = st.sc;
  WriteIop& io = st.io;
  Group &g_accum = st.g_accum, &g_code = st.g_code, &g_data = st.g_data, &g_check = st.g_check;
  Group* grp[3] = {&g_accum, &g_code, &g_data};
  const std::vector<uint32_t>& mix = st.mix;
  const uint32_t* global = st.global.data();
  if (!g_data.evaluated) {  // r0h_proof_shrink gave the evaluations back: the same expanding NTT over the kept coefficients
    phase(ctx, "evaluate_data_again");
    R0H_TRY(sc.alloc(ctx, ((size_t)g_data.count << (po2 + 2)) * 4, &g_data.evaluated));
    R0H_TRY(r0h_batch_expand_into_evaluate_ntt(ctx, g_data.evaluated, g_data.coeffs, g_data.count, po2, 2));
    g_data.tree.matrix = g_data.evaluated;
  }
  phase(ctx, "commit_accum");
  R0H_TRY(group_from_witness(ctx, sc, g_accum, accum, po2));
  R0H_TRY(tree_commit(ctx, g_accum.tree, io));

  phase(ctx, "eval_check");
  Fp4 poly_mix = io.rng.ext();
  R0H_TRY(sc.alloc(ctx, domain * 16, &g_check.coeffs));
  R0H_TRY(r0h_eval_check(ctx, circ, po2, g_accum.evaluated, g_code.evaluated, g_data.evaluated, global, mix.data(), poly_mix.e, g_check.coeffs));
  phase(ctx, "commit_check");
  R0H_TRY(r0h_batch_interpolate_ntt(ctx, g_check.coeffs, 4, po2 + 2));  // 4 polys of 4N == 16 polys of N (bit-reversed)
  R0H_TRY(group_finish(ctx, sc, g_check, po2));
  R0H_TRY(tree_commit(ctx, g_check.tree, io));

  phase(ctx, "evaluate_at_z");
  const Fp4 z = io.rng.ext(), z4 = fp4_pow(z, 4);
  const uint32_t back_one = rou_rev(po2);
  const uint32_t n_u = cv.n_taps + R0H_CHECK_SIZE;
  std::vector<Fp4> all_xs(n_u), coeff_u(n_u);
  std::vector<uint32_t> which(n_u);
  r0h_buf* d_eval = nullptr;
  R0H_TRY(sc.alloc(ctx, (size_t)n_u * 16, &d_eval));
  for (uint32_t t = 0; t < cv.n_taps; t++) {
    all_xs[t] = scale(z, fpow(back_one, cv.tap_back[t]));
    which[t] = cv.tap_offset[t];
  }
  for (uint32_t i = 0; i < R0H_CHECK_SIZE; i++) { all_xs[cv.n_taps + i] = z4; which[cv.n_taps + i] = i; }
  for (int g = 0; g < 3; g++) {
    uint32_t b = cv.group_tap_begin[g], e = cv.group_tap_begin[g + 1];
    if (e == b) continue;
    r0h_buf view = *d_eval;
    view.ptr = (char*)d_eval->ptr + (size_t)b * 16;
    view.bytes = (size_t)(e - b) * 16;
    R0H_TRY(evaluate_any(ctx, grp[g]->coeffs, po2, which.data() + b, (const uint32_t*)(all_xs.data() + b), e - b, &view, true));
  }
  {
    r0h_buf view = *d_eval;
    view.ptr = (char*)d_eval->ptr + (size_t)cv.n_taps * 16;
    view.bytes = (size_t)R0H_CHECK_SIZE * 16;
    R0H_TRY(evaluate_any(ctx, g_check.coeffs, po2, which.data() + cv.n_taps, (const uint32_t*)(all_xs.data() + cv.n_taps), R0H_CHECK_SIZE, &view, true));
  }
  std::vector<Fp4> eval_u(n_u);
  R0H_TRY(r0h_buf_d2h(ctx, d_eval, 0, eval_u.data(), (size_t)n_u * 16));
  for (uint32_t r = 0; r < cv.n_regs; r++) {
    uint32_t p = cv.reg_first[r];
    poly_interpolate(&coeff_u[p], &all_xs[p], &eval_u[p], cv.reg_size[r]);
  }
  for (uint32_t i = 0; i < R0H_CHECK_SIZE; i++) coeff_u[cv.n_taps + i] = eval_u[cv.n_taps + i];
  io.write((const uint32_t*)coeff_u.data(), 4 * (size_t)n_u);
  {
    uint32_t d[8];
    p2_hash_elems_host(ctx->p2_host, (const uint32_t*)coeff_u.data(), 4 * (size_t)n_u, d);
    io.commit(d);
  }

  phase(ctx, "mix_combos");
  const Fp4 mixv = io.rng.ext();
  const uint32_t n_combos = cv.n_combos;
  r0h_buf* combos = nullptr;
  R0H_TRY(sc.alloc(ctx, (size_t)(n_combos + 1) * n * 16, &combos));
  R0H_TRY(r0h_buf_zero(ctx, combos));
  Fp4 cur = fp4_one();
  for (int g = 0; g < 3; g++) {
    uint32_t gs = cv.group_size[g];
    std::vector<uint32_t> combo_of(gs);
    for (uint32_t r = 0; r < cv.n_regs; r++)
      if (cv.reg_group[r] == (uint32_t)g) combo_of[cv.reg_offset[r]] = cv.reg_combo[r];
    R0H_TRY(r0h_mix_poly_coeffs(ctx, combos, cur.e, mixv.e, grp[g]->coeffs, combo_of.data(), gs, po2));
    cur = cur * fp4_pow(mixv, gs);
  }
  {
    uint32_t combo_of[R0H_CHECK_SIZE];
    for (int i = 0; i < R0H_CHECK_SIZE; i++) combo_of[i] = n_combos;
    R0H_TRY(r0h_mix_poly_coeffs(ctx, combos, cur.e, mixv.e, g_check.coeffs, combo_of, R0H_CHECK_SIZE, po2));
  }
  R0H_TRY(bit_reverse_ext(ctx, combos, n_combos + 1, po2));  // combos were mixed in bit-reversed order: natural from here on
  // subtract the interpolants: only the first few coefficients of each combo change
  {
    std::vector<Fp4> head((size_t)(n_combos + 1) * 64, fp4_zero());
    std::vector<uint32_t> head_len(n_combos + 1, 0);
    cur = fp4_one();
    for (uint32_t r = 0; r < cv.n_regs; r++) {
      R0H_REQUIRE(cv.reg_size[r] <= 64, "prove_segment: register with more than 64 taps");
      for (uint32_t i = 0; i < cv.reg_size[r]; i++) {
        Fp4& h = head[(size_t)cv.reg_combo[r] * 64 + i];
        h = h + cur * coeff_u[cv.reg_first[r] + i];
      }
      if (cv.reg_size[r] > head_len[cv.reg_combo[r]]) head_len[cv.reg_combo[r]] = cv.reg_size[r];
      cur = cur * mixv;
    }
    for (uint32_t i = 0; i < R0H_CHECK_SIZE; i++) {
      head[(size_t)n_combos * 64] = head[(size_t)n_combos * 64] + cur * coeff_u[cv.n_taps + i];
      cur = cur * mixv;
    }
    head_len[n_combos] = 1;
    std::vector<uint32_t> fix;
    for (uint32_t k = 0; k <= n_combos; k++)
      for (uint32_t i = 0; i < head_len[k]; i++)
        for (uint32_t q = 0; q < 4; q++) {
          fix.push_back((uint32_t)((((size_t)k << po2) + i) * 4 + q));
          fix.push_back(head[(size_t)k * 64 + i].e[q]);
        }
    R0H_REQUIRE(((size_t)(n_combos + 1) << po2) * 4 < ((size_t)1 << 32), "prove_segment: combos buffer exceeds 32-bit word indexing");
    r0h_buf* d_fix = nullptr;
    R0H_TRY(sc.alloc(ctx, fix.size() * 4, &d_fix));
    R0H_TRY(stage_h2d(ctx, d_fix->ptr, fix.data(), fix.size() * 4));
    uint32_t nf = (uint32_t)(fix.size() / 2);
    hipLaunchKernelGGL(sub_head_kernel, dim3((nf + 255) / 256), dim3(256), 0, ctx->stream, u32(combos), u32(d_fix), nf);
    {
      hipError_t e = hipGetLastError();
      R0H_REQUIRE(e == hipSuccess, "sub_head_kernel: %s", hipGetErrorString(e));
    }
    sc.release(d_fix);
  }
  phase(ctx, "deep_divide");
  {
    // every (combo, point) division of the DEEP step as one batch: a launch set per "k-th point of every combo", one read-back
    // of all remainders (upstream divides on the host; round 1 here ran one scan and one host sync per division)
    std::vector<uint32_t> job_poly, job_pt;
    for (uint32_t k = 0; k <= n_combos; k++) {
      if (k < n_combos) {
        for (uint32_t b = cv.combo_begin[k]; b < cv.combo_begin[k + 1]; b++) {
          const Fp4 pt = scale(z, fpow(back_one, cv.combo_backs[b]));
          job_poly.push_back(k);
          job_pt.insert(job_pt.end(), pt.e, pt.e + 4);
        }
      } else {
        job_poly.push_back(k);
        job_pt.insert(job_pt.end(), z4.e, z4.e + 4);
      }
    }
    std::vector<uint32_t> rem(4 * job_poly.size());
    R0H_TRY(poly_divide_batch(ctx, combos, (uint32_t)n, job_poly.data(), job_pt.data(), (uint32_t)job_poly.size(), rem.data()));
    for (size_t j = 0; j < job_poly.size(); j++)
      R0H_REQUIRE(!(rem[4 * j] | rem[4 * j + 1] | rem[4 * j + 2] | rem[4 * j + 3]),
                  "prove_segment: DEEP quotient of combo %u has a non-zero remainder (witness violates the taps?)", job_poly[j]);
  }
  r0h_buf* fri_coeffs = nullptr;
  R0H_TRY(sc.alloc(ctx, n * 16, &fri_coeffs));
  R0H_TRY(r0h_eltwise_sum_extelem(ctx, fri_coeffs, combos, n_combos + 1, (uint32_t)n));
  R0H_TRY(r0h_batch_bit_reverse(ctx, fri_coeffs, 4, po2));
  sc.release(combos);

  // ---- FRI
  phase(ctx, "fri_commit");
  struct Round { Tree tree; r0h_buf* evaluated; size_t domain; };
  std::vector<Round> rounds;
  size_t deg = n;
  while (deg > R0H_FRI_MIN_DEGREE) {
    Round rd{Tree(deg * R0H_INV_RATE / R0H_FRI_FOLD, R0H_FRI_FOLD * 4), nullptr, deg * R0H_INV_RATE};
    R0H_TRY(sc.alloc(ctx, rd.domain * 16, &rd.evaluated));
    R0H_TRY(r0h_batch_expand_into_evaluate_ntt(ctx, rd.evaluated, fri_coeffs, 4, log2u(deg), 2));
    R0H_TRY(tree_build(ctx, sc, rd.tree, rd.evaluated));
    R0H_TRY(tree_commit(ctx, rd.tree, io));
    Fp4 fold_mix = io.rng.ext();
    r0h_buf* folded = nullptr;
    R0H_TRY(sc.alloc(ctx, deg / R0H_FRI_FOLD * 16, &folded));
    R0H_TRY(r0h_fri_fold(ctx, folded, fri_coeffs, fold_mix.e, (uint32_t)(deg / R0H_FRI_FOLD)));
    sc.release(fri_coeffs);
    fri_coeffs = folded;
    deg /= R0H_FRI_FOLD;
    rounds.push_back(rd);
  }
  R0H_TRY(r0h_batch_bit_reverse(ctx, fri_coeffs, 4, log2u(deg)));
  {
    std::vector<uint32_t> fc(4 * deg);
    R0H_TRY(r0h_buf_d2h(ctx, fri_coeffs, 0, fc.data(), fc.size() * 4));
    io.write(fc.data(), fc.size());
    uint32_t d[8];
    p2_hash_elems_host(ctx->p2_host, fc.data(), fc.size(), d);
    io.commit(d);
  }

  phase(ctx, "queries");
  {
    // the query positions depend only on the transcript state, not on the openings: draw all of them, then open each
    // tree for all queries in one kernel + one copy
    const uint32_t nq = R0H_QUERIES, n_trees = 4 + (uint32_t)rounds.size();
    std::vector<uint32_t> idx((size_t)n_trees * nq);
    for (uint32_t q = 0; q < nq; q++) {
      size_t pos = io.rng.bits(log2u(domain)) % domain;
      for (uint32_t t = 0; t < 4; t++) idx[(size_t)t * nq + q] = (uint32_t)pos;
      for (size_t r = 0; r < rounds.size(); r++) {
        pos = pos % (rounds[r].domain / R0H_FRI_FOLD);
        idx[(size_t)(4 + r) * nq + q] = (uint32_t)pos;
      }
    }
    r0h_buf* d_idx = nullptr;
    R0H_TRY(sc.alloc(ctx, idx.size() * 4, &d_idx));
    R0H_TRY(stage_h2d(ctx, d_idx->ptr, idx.data(), idx.size() * 4));
    std::vector<const Tree*> trees = {&g_accum.tree, &g_code.tree, &g_data.tree, &g_check.tree};
    for (Round& rd : rounds) trees.push_back(&rd.tree);
    std::vector<std::vector<uint32_t>> opened(n_trees);
    for (uint32_t t = 0; t < n_trees; t++) {
      r0h_buf view = *d_idx;
      view.ptr = (char*)d_idx->ptr + (size_t)t * nq * 4;
      view.bytes = (size_t)nq * 4;
      R0H_TRY(tree_open(ctx, sc, *trees[t], &view, nq, opened[t]));
    }
    for (uint32_t q = 0; q < nq; q++)
      for (uint32_t t = 0; t < n_trees; t++) {
        size_t w = trees[t]->mp.opening_words();
        io.write(opened[t].data() + (size_t)q * w, w);
      }
  }
  phase(ctx, "end");
  R0H_TRY_HIP(hipStreamSynchronize(ctx->stream));
  Profile& pf = ctx->prof;
  pf.ms.assign(pf.names.size(), 0.f);
  for (size_t i = 0; i + 1 < pf.names.size(); i++) hipEventElapsedTime(&pf.ms[i], pf.events[i], pf.events[i + 1]);
  seal.swap(io.proof);
  return nullptr;
}

}  // namespace r0h

using namespace r0h;

extern "C" {

static const char* prove_segment_impl(r0h_ctx* ctx, const r0h_circuit* c, uint32_t po2, const r0h_buf* code, const r0h_code_commit* cc,
                                      const r0h_buf* data, const uint32_t* global, uint32_t* seal_out, size_t seal_cap, size_t* seal_words_out) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(ctx && c && (code || cc) && data && seal_words_out, "r0h_prove_segment: NULL argument");
  R0H_REQUIRE(global || r0h_circuit_n_global(c) == 0, "r0h_prove_segment: global is NULL");
  R0H_REQUIRE(po2 >= 9 && po2 <= R0H_MAX_PO2, "r0h_prove_segment: po2 %u outside [9, %u]", po2, R0H_MAX_PO2);
  R0H_REQUIRE(c->has_column_program, "r0h_prove_segment: the circuit has no accumulation program; drive the per-op entry points with your own accum step");
  R0H_TRY_HIP(hipSetDevice(ctx->device));
  std::vector<uint32_t> seal;
  {
    CircuitView cv;
    circuit_view(c, &cv);
    r0h_proof st(ctx, c, po2, cv);
    R0H_TRY(proof_begin(st, code, cc, data, global));
    if (c->n_late) R0H_TRY(proof_late(st, global + (cv.n_global - c->n_late)));
    r0h_buf* accum = nullptr;
    R0H_TRY(st.sc.alloc(ctx, ((size_t)cv.group_size[R0H_GROUP_ACCUM] << po2) * 4, &accum));
    const r0h_buf* code_cols = code;
    r0h_buf code_view;
    if (!code_cols && cc && cc->witness) { code_view = *cc->witness; code_cols = &code_view; }
    R0H_TRY(r0h_accum_public(ctx, c, po2, code_cols, data, st.global.data(), st.mix.data(), accum));
    R0H_TRY(proof_finish(st, accum, seal));
  }
  *seal_words_out = seal.size();
  R0H_REQUIRE(seal.size() <= seal_cap || !seal_out, "r0h_prove_segment: seal needs %zu words, capacity is %zu", seal.size(), seal_cap);
  if (seal_out) memcpy(seal_out, seal.data(), seal.size() * 4);
  return nullptr;
  R0H_GUARD_END
}

static const char* proof_begin_impl(r0h_ctx* ctx, const r0h_circuit* c, uint32_t po2, const r0h_buf* code, const r0h_code_commit* cc,
                                    const r0h_buf* data, const uint32_t* global, uint32_t* mix_out, r0h_proof** out) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(ctx && c && (code || cc) && data && out, "r0h_proof_begin: NULL argument");
  R0H_REQUIRE((global || r0h_circuit_n_global(c) == 0) && (mix_out || r0h_circuit_n_mix(c) == 0 || c->n_late), "r0h_proof_begin: NULL globals / mix_out");
  R0H_REQUIRE(po2 >= 9 && po2 <= R0H_MAX_PO2, "r0h_proof_begin: po2 %u outside [9, %u]", po2, R0H_MAX_PO2);
  R0H_TRY_HIP(hipSetDevice(ctx->device));
  CircuitView cv;
  circuit_view(c, &cv);
  r0h_proof* st = new r0h_proof(ctx, c, po2, cv);
  const char* err = proof_begin(*st, code, cc, data, global);
  if (err) { delete st; return err; }
  if (cv.n_mix && st->mix_drawn && mix_out) memcpy(mix_out, st->mix.data(), (size_t)cv.n_mix * 4);
  *out = st;
  return nullptr;
  R0H_GUARD_END
}

const char* r0h_prove_segment(r0h_ctx* ctx, const r0h_circuit* c, uint32_t po2, const r0h_buf* code, const r0h_buf* data,
                              const uint32_t* global, uint32_t* seal_out, size_t seal_cap, size_t* seal_words_out) {
  R0H_REQUIRE(code, "r0h_prove_segment: NULL argument");
  return prove_segment_impl(ctx, c, po2, code, nullptr, data, global, seal_out, seal_cap, seal_words_out);
}
const char* r0h_prove_segment_committed(r0h_ctx* ctx, const r0h_circuit* c, uint32_t po2, const r0h_code_commit* code, const r0h_buf* data,
                                        const uint32_t* global, uint32_t* seal_out, size_t seal_cap, size_t* seal_words_out) {
  R0H_REQUIRE(code, "r0h_prove_segment_committed: NULL argument");
  return prove_segment_impl(ctx, c, po2, nullptr, code, data, global, seal_out, seal_cap, seal_words_out);
}

const char* r0h_proof_begin(r0h_ctx* ctx, const r0h_circuit* c, uint32_t po2, const r0h_buf* code, const r0h_buf* data,
                            const uint32_t* global, uint32_t* mix_out, r0h_proof** out) {
  R0H_REQUIRE(code, "r0h_proof_begin: NULL argument");
  return proof_begin_impl(ctx, c, po2, code, nullptr, data, global, mix_out, out);
}
const char* r0h_proof_begin_committed(r0h_ctx* ctx, const r0h_circuit* c, uint32_t po2, const r0h_code_commit* code, const r0h_buf* data,
                                      const uint32_t* global, uint32_t* mix_out, r0h_proof** out) {
  R0H_REQUIRE(code, "r0h_proof_begin_committed: NULL argument");
  return proof_begin_impl(ctx, c, po2, nullptr, code, data, global, mix_out, out);
}

// Commit the CODE group once (same steps as the sequencer's commit_code) and keep it: see r0h_code_commit above.
const char* r0h_code_commit_new(r0h_ctx* ctx, const r0h_buf* code, uint32_t count, uint32_t po2, r0h_code_commit** out) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(ctx && code && out, "r0h_code_commit_new: NULL argument");
  R0H_REQUIRE(count >= 1, "r0h_code_commit_new: the CODE group has no columns");
  R0H_REQUIRE(po2 >= 9 && po2 <= R0H_MAX_PO2, "r0h_code_commit_new: po2 %u outside [9, %u]", po2, R0H_MAX_PO2);
  R0H_TRY_HIP(hipSetDevice(ctx->device));
  std::unique_ptr<r0h_code_commit, const char* (*)(r0h_code_commit*)> cc(new r0h_code_commit(), r0h_code_commit_free);
  cc->ctx = ctx; cc->count = count; cc->po2 = po2;
  ctx_retain(ctx);
  {
    Scope sc;  // the group's buffers are taken out of the scope once everything has succeeded
    Group g(count, (size_t)4 << po2);
    R0H_TRY(group_from_witness(ctx, sc, g, code, po2));
    WriteIop io(&ctx->p2_host);
    R0H_TRY(tree_commit(ctx, g.tree, io));  // blocking read-back: the stream has drained when it returns
    cc->top.swap(io.proof);
    R0H_TRY(r0h_buf_d2h(ctx, g.tree.nodes, 32, cc->root, 32));
    cc->coeffs = g.coeffs; cc->evaluated = g.evaluated; cc->nodes = g.tree.nodes;
    {  // a copy of the columns stays with the commitment
      const size_t bytes = ((size_t)count << po2) * 4;
      R0H_TRY(r0h_buf_alloc(ctx, bytes, &cc->witness));
      R0H_TRY_HIP(hipMemcpyAsync(cc->witness->ptr, code->ptr, bytes, hipMemcpyDeviceToDevice, ctx->stream));
      R0H_TRY_HIP(hipStreamSynchronize(ctx->stream));
    }
    sc.bufs.clear();
    // long-lived and read from other contexts' streams: on release these go back to the device (hipFree waits for every stream),
    // not into this context's stream-ordered pool
    for (r0h_buf* b : {cc->coeffs, cc->evaluated, cc->nodes}) { b->pooled = false; b->owned = true; }
  }
  *out = cc.release();
  return nullptr;
  R0H_GUARD_END
}
const char* r0h_code_commit_free(r0h_code_commit* cc) {
  if (!cc) return nullptr;
  // a proof on another context may still be reading these buffers on its own stream: the caller frees a commitment only after
  // the proofs that use it have returned (r0h_prove_segment_committed / r0h_proof_finish block until the seal is out)
  if (cc->coeffs) r0h_buf_free(cc->coeffs);
  if (cc->evaluated) r0h_buf_free(cc->evaluated);
  if (cc->nodes) r0h_buf_free(cc->nodes);
  if (cc->witness) r0h_buf_free(cc->witness);
  r0h_ctx* ctx = cc->ctx;
  delete cc;
  if (ctx) ctx_release(ctx);
  return nullptr;
}
const char* r0h_code_commit_columns(const r0h_code_commit* cc, const r0h_buf** columns_out) {
  R0H_REQUIRE(cc && columns_out && cc->witness, "r0h_code_commit_columns: NULL argument");
  *columns_out = cc->witness;
  return nullptr;
}
const char* r0h_code_commit_root(const r0h_code_commit* cc, uint32_t root_out[8]) {
  R0H_REQUIRE(cc && root_out, "r0h_code_commit_root: NULL argument");
  memcpy(root_out, cc->root, 32);
  return nullptr;
}

const char* r0h_proof_finish(r0h_proof* proof, const r0h_buf* accum, uint32_t* seal_out, size_t seal_cap, size_t* seal_words_out) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(proof && accum && seal_words_out, "r0h_proof_finish: NULL argument");
  R0H_TRY_HIP(hipSetDevice(proof->ctx->device));
  std::vector<uint32_t> seal;
  const char* err = proof_finish(*proof, accum, seal);
  delete proof;  // consumed either way
  if (err) return err;
  *seal_words_out = seal.size();
  R0H_REQUIRE(seal.size() <= seal_cap || !seal_out, "r0h_proof_finish: seal needs %zu words, capacity is %zu", seal.size(), seal_cap);
  if (seal_out) memcpy(seal_out, seal.data(), seal.size() * 4);
  return nullptr;
  R0H_GUARD_END
}

const char* r0h_proof_data_root(const r0h_proof* proof, uint32_t root_out[8]) {
  R0H_REQUIRE(proof && root_out, "r0h_proof_data_root: NULL argument");
  memcpy(root_out, proof->data_root, 32);
  return nullptr;
}
const char* r0h_proof_late(r0h_proof* proof, const uint32_t* late_globals, uint32_t* mix_out) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(proof && (mix_out || !proof->cv.n_mix), "r0h_proof_late: NULL argument");
  R0H_REQUIRE(proof->circ->n_late, "r0h_proof_late: this circuit has no late public inputs");
  R0H_TRY_HIP(hipSetDevice(proof->ctx->device));
  R0H_TRY(proof_late(*proof, late_globals));
  if (proof->cv.n_mix) memcpy(mix_out, proof->mix.data(), (size_t)proof->cv.n_mix * 4);
  return nullptr;
  R0H_GUARD_END
}
const char* r0h_proof_globals(const r0h_proof* proof, uint32_t* globals_out) {
  R0H_REQUIRE(proof && globals_out, "r0h_proof_globals: NULL argument");
  memcpy(globals_out, proof->global.data(), proof->global.size() * 4);
  return nullptr;
}

const char* r0h_proof_abort(r0h_proof* proof) {
  delete proof;
  return nullptr;
}
// Give back the DATA group's evaluations on the 4N coset (4/5 of what a proof holds between its phases): the commitment -- the Merkle
// nodes -- and the coefficients stay, r0h_proof_finish evaluates again (the same words: one expanding NTT).
const char* r0h_proof_shrink(r0h_proof* proof, size_t* bytes_freed_out) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(proof, "r0h_proof_shrink: NULL argument");
  size_t freed = 0;
  if (proof->g_data.evaluated) {
    R0H_TRY_HIP(hipSetDevice(proof->ctx->device));
    R0H_TRY_HIP(hipStreamSynchronize(proof->ctx->stream));  // the block goes back to the pool: nothing in flight may still read it
    freed = proof->g_data.evaluated->bytes;
    proof->sc.release(proof->g_data.evaluated);
    proof->g_data.evaluated = nullptr;
    proof->g_data.tree.matrix = nullptr;
  }
  if (bytes_freed_out) *bytes_freed_out = freed;
  return nullptr;
  R0H_GUARD_END
}
size_t r0h_proof_resident_bytes(const r0h_proof* proof) {
  if (!proof) return 0;
  size_t total = 0;
  for (const r0h_buf* b : proof->sc.bufs) total += b->bytes;
  return total;
}

// The control root of a program at one trace size: the Merkle root of its committed CODE group (risc0 keeps one such root per
// po2 -- the control id -- and `verify` compares the seal's CODE commitment with it).  Same steps as the sequencer's commit_code.
const char* r0h_code_root(r0h_ctx* ctx, const r0h_buf* code, uint32_t count, uint32_t po2, uint32_t root_out[8]) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(ctx && code && root_out, "r0h_code_root: NULL argument");
  R0H_REQUIRE(count >= 1, "r0h_code_root: the CODE group has no columns");
  R0H_REQUIRE(po2 >= 9 && po2 <= R0H_MAX_PO2, "r0h_code_root: po2 %u outside [9, %u]", po2, R0H_MAX_PO2);
  R0H_TRY_HIP(hipSetDevice(ctx->device));
  Scope sc;
  Group g(count, (size_t)4 << po2);
  R0H_TRY(group_from_witness(ctx, sc, g, code, po2));
  return r0h_buf_d2h(ctx, g.tree.nodes, 32, root_out, 32);
  R0H_GUARD_END
}

const char* r0h_last_profile(r0h_ctx* ctx, const char*** names_out, const float** ms_out, uint32_t* n_out) {
  R0H_REQUIRE(ctx && names_out && ms_out && n_out, "r0h_last_profile: NULL argument");
  *names_out = ctx->prof.names.data();
  *ms_out = ctx->prof.ms.data();
  *n_out = ctx->prof.names.empty() ? 0 : (uint32_t)ctx->prof.names.size() - 1;
  return nullptr;
}

}  // extern "C"
