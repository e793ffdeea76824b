// `Prover::prove(env, elf)` in the shape hyperfridge calls it (host/src/main.rs:420-423): execute the guest, cut the run
// into segments, prove every segment, assemble the composite receipt.  risc0-zkvm 3.0.5 `LocalProver::prove` = executor
// (`ExecutorImpl::run` -> Session{segments}) + `prove_session` (one `prove_segment` per segment) + CompositeReceipt.
//
// What each stage is here: the executor is csrc/rv32im.cpp (RV32IM by the specification; this library's ecall ABI and cycle model,
// see there); `prove_segment` is the device-resident sequencer (csrc/prover.hip) at the trace size the segment needs; the claim of
// every segment (system states from the run, exit code, the journal's output digest on the last one) is bound to its seal through
// the public inputs (csrc/claim.cpp).
//
// With the trace circuit (circuits/trace.r0c, ProtocolInfo "R0HIP_TRACE:v5__") the seal of a segment attests THAT segment: the
// executor keeps one compact row per cycle plus one per address touched, the device expands them into the DATA group
// (csrc/trace.hip: 72 bytes per cycle cross PCIe, not the 552 bytes of an expanded row), and the proof is over those columns --
// contiguity, every instruction's semantics and memory consistency as include/r0hip.h lists them.  The guest runs ahead on its
// own host thread; two prover lanes (the caller's context and a helper context of the same device) take the segments as they
// are cut.
//
// A session is proved in TWO PHASES, because its segments are tied together by one argument (the session-wide memory
// argument): phase 1 commits the DATA group of every segment and keeps the proofs in flight (288 GB of HBM hold dozens of 3 GB
// commitments); the session challenge is derived from all the DATA roots (and the segments' early public inputs); phase 2 gives
// every proof its late public inputs -- the challenge and the segment's sum under it -- and finishes it.  Ranks that share a
// session exchange 28 words per segment between the phases (r0h_session_begin / _records / _finish; one rank: r0h_prove_elf).
// With any other circuit the witness is the blob's synthetic column program with the claim planted: the seal then proves "a
// satisfying trace of the loaded circuit exists whose public inputs name this claim", not "this program ran".
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "circuit.hpp"
#include "receipt_types.hpp"

using namespace r0h;

namespace {
using Clock = std::chrono::steady_clock;
double seconds(Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double>(b - a).count(); }

struct CodeCommits {  // one committed CODE group per trace size met in the run
  std::map<uint32_t, r0h_code_commit*> by_po2;
  ~CodeCommits() { for (auto& kv : by_po2) r0h_code_commit_free(kv.second); }
  const char* get(r0h_ctx* ctx, const r0h_circuit* c, uint32_t po2, r0h_buf* data_scratch, r0h_code_commit** out) {
    auto it = by_po2.find(po2);
    if (it == by_po2.end()) {
      // the fixed CODE columns come from the blob's column program (the DATA it also fills is scratch here)
      const size_t n = (size_t)1 << po2;
      r0h_buf* code = nullptr;
      R0H_TRY(buf_alloc_pooled(ctx, (size_t)c->group_size[R0H_GROUP_CODE] * n * 4, &code));
      const char* err = r0h_witgen(ctx, c, po2, 0, code, data_scratch, nullptr);
      r0h_code_commit* cc = nullptr;
      if (!err) err = r0h_code_commit_new(ctx, code, c->group_size[R0H_GROUP_CODE], po2, &cc);
      r0h_buf_free(code);
      if (err) return err;
      it = by_po2.emplace(po2, cc).first;
    }
    *out = it->second;
    return nullptr;
  }
};

struct Produced {  // a segment as the executor thread hands it over
  size_t index = 0;
  r0h_vm_segment info;
  r0h_receipt_claim claim;
  std::vector<r0h_preflight_row> rows;
  std::vector<r0h_preflight_bound> bounds;
};

uint32_t trace_size(uint64_t rows) {
  uint32_t po2 = 9;  // the sequencer's smallest trace
  while (((uint64_t)1 << po2) < rows) po2++;
  return po2;
}

// Row buffers a context keeps between sessions (r0h_prove_elf): vectors with their heap blocks, some of them page-locked.
struct RowPool {
  std::mutex mu;
  std::vector<std::vector<r0h_preflight_row>> rows;
  std::vector<std::vector<r0h_preflight_bound>> bounds;
  std::map<const void*, size_t> pinned;
  void unpin(const void* p) {
    auto it = pinned.find(p);
    if (it == pinned.end()) return;
    (void)hipHostUnregister(const_cast<void*>(p));
    pinned.erase(it);
  }
  // keep at most `cap` row buffers (pinned ones first); a pinned block that is not among them is unpinned BEFORE its vector is freed
  void settle(size_t cap) {
    for (size_t k = 0; k < rows.size();)
      if (!rows[k].capacity()) rows.erase(rows.begin() + k);
      else k++;
    std::stable_sort(rows.begin(), rows.end(), [&](const std::vector<r0h_preflight_row>& a, const std::vector<r0h_preflight_row>& b) {
      return (pinned.count(a.data()) != 0) > (pinned.count(b.data()) != 0);
    });
    const size_t keep = std::min(cap, rows.size());
    for (auto it = pinned.begin(); it != pinned.end();) {
      bool alive = false;
      for (size_t k = 0; k < keep; k++) alive = alive || rows[k].data() == it->first;
      if (alive) { ++it; continue; }
      (void)hipHostUnregister(const_cast<void*>(it->first));
      it = pinned.erase(it);
    }
    rows.resize(keep);
    if (bounds.size() > cap) bounds.resize(cap);
  }
};
RowPool* pool_of(r0h_ctx* ctx) {
  if (!ctx->session_rows) ctx->session_rows = new RowPool();
  return (RowPool*)ctx->session_rows;
}
}  // namespace

namespace r0h {
void session_rows_free(r0h_ctx* ctx) {
  RowPool* p = (RowPool*)ctx->session_rows;
  if (!p) return;
  p->settle(0);  // unpins every block, then frees them
  delete p;
  ctx->session_rows = nullptr;
}
}  // namespace r0h


// the record a segment contributes to the session challenge: its early public inputs, then the root of its DATA commitment
constexpr uint32_t RECORD_WORDS = (R0H_TRACE_GLOBALS - R0H_TRACE_LATE_GLOBALS) + 8;
static_assert(RECORD_WORDS == R0H_SESSION_RECORD_WORDS, "R0H_SESSION_RECORD_WORDS out of step");

struct Pending {  // a segment between the phases: committed, waiting for the session challenge
  size_t index = 0;
  r0h_ctx* lctx = nullptr;
  r0h_proof* proof = nullptr;
  r0h_buf* data = nullptr;
  r0h_code_commit* cc = nullptr;
  uint32_t po2 = 0;
  std::vector<uint32_t> global;
  uint32_t root[8] = {0};
  r0h_receipt_claim claim;
  std::vector<uint32_t> seal;  // other circuits: proved at once
  bool done = false;
};

struct r0h_session {
  r0h_ctx* ctx = nullptr;
  const r0h_circuit* c = nullptr;
  bool trace_mode = false;
  uint32_t part = 0, parts = 1;
  size_t n_segments = 0;
  std::vector<Pending> pending;  // this rank's segments, by index
  CodeCommits commits;
  std::vector<uint8_t> journal;
  std::vector<uint8_t> elf;  // kept for the image proof (r0h_ctx_set_image_circuit)
  uint8_t image_id[32] = {0};
  uint64_t cycles = 0;
  r0h_session_stats stats = {0, 0, 0, 0, 0, 0, 0};
  uint64_t resident_limit = 0, resident_evaluations = 0;  // bytes of DATA evaluations kept between the phases (r0h_ctx_set_session_resident_limit)
  Clock::time_point t_begin;
  std::vector<r0h_ctx*> lane_ctx;
  ~r0h_session() {
    for (Pending& p : pending) {
      if (p.proof) r0h_proof_abort(p.proof);
      if (p.data) r0h_buf_free(p.data);
    }
    if (ctx) ctx_release(ctx);
  }
};

extern "C" {

const char* r0h_last_session_stats(r0h_ctx* ctx, r0h_session_stats* out) {
  R0H_REQUIRE(ctx && out, "r0h_last_session_stats: NULL argument");
  *out = ctx->session;
  return nullptr;
}

const char* r0h_session_free(r0h_session* s) {
  delete s;
  return nullptr;
}

// Phase 1: execute the guest, cut it into segments, and -- for this rank's share of them -- expand the rows on the device and commit
// the DATA group (trace circuit), or prove outright (any other circuit).
const char* r0h_session_begin(r0h_ctx* ctx, const r0h_circuit* c, const uint8_t* elf, size_t elf_len, const uint32_t* input_words, size_t n_input, uint32_t segment_po2,
                              uint64_t max_cycles, uint32_t part, uint32_t parts, r0h_session** session_out) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(ctx && c && elf && session_out && (input_words || !n_input), "r0h_prove_elf: NULL argument");
  R0H_REQUIRE(parts >= 1 && part < parts, "r0h_prove_elf_part: part %u of %u", part, parts);
  const bool trace_mode = !memcmp(c->info, "R0HIP_TRACE:v5__", 16);
  // (a trace-circuit segment of fewer than 2^16 rows is proved at 2^16: the lookup tables' size)
  R0H_REQUIRE(segment_po2 >= 9 && segment_po2 <= (trace_mode ? R0H_TRACE_MAX_PO2 : R0H_MAX_PO2), "r0h_prove_elf: segment_po2 %u outside [9, %u]", segment_po2,
              (unsigned)(trace_mode ? R0H_TRACE_MAX_PO2 : R0H_MAX_PO2));
  R0H_REQUIRE(c->has_column_program, "r0h_prove_elf: the circuit has no column program (CODE columns, accumulation)");
  R0H_REQUIRE(c->n_global >= 8, "r0h_prove_elf: the circuit exposes %u public inputs, a claim needs 8", c->n_global);
  if (trace_mode)
    R0H_REQUIRE(c->n_global == R0H_TRACE_GLOBALS && c->n_late == R0H_TRACE_LATE_GLOBALS && c->group_size[R0H_GROUP_DATA] == R0H_TRACE_COLUMNS,
                "r0h_prove_elf: this trace circuit is not the one the library was built for");
  else
    R0H_REQUIRE(c->n_late == 0, "r0h_prove_elf: a circuit with late public inputs other than the trace circuit");
  if (!max_cycles) max_cycles = R0H_DEFAULT_SESSION_LIMIT;  // a guest that never halts must not hang the host (risc0: session limit)
  std::unique_ptr<r0h_session> ses(new r0h_session());
  ses->ctx = ctx;
  ctx_retain(ctx);
  ses->c = c;
  ses->trace_mode = trace_mode;
  ses->part = part;
  ses->parts = parts;
  ses->t_begin = Clock::now();
  if (trace_mode && ctx->image_circuit && part == 0) ses->elf.assign(elf, elf + elf_len);
  ses->resident_limit = ctx->session_resident_limit;
  if (!ses->resident_limit) {
    size_t free_b = 0, total_b = 0;
    R0H_TRY_HIP(hipSetDevice(ctx->device));
    R0H_TRY_HIP(hipMemGetInfo(&free_b, &total_b));
    ses->resident_limit = total_b / 8;  // 36 GB of 288: 17 segments of 2^20 rows; the camt53 stand-in's 12 stay whole, the reference's 37 do not
  }
  r0h_session_stats& stats = ses->stats;
  r0h_vm* vm = nullptr;
  R0H_TRY(r0h_vm_new(&vm));
  struct VmGuard { r0h_vm* v; ~VmGuard() { r0h_vm_free(v); } } guard{vm};
  R0H_TRY(r0h_vm_load_elf(vm, elf, elf_len));
  R0H_TRY(r0h_vm_set_input(vm, input_words, n_input));
  r0h_vm_limits lim;
  memset(&lim, 0, sizeof lim);
  lim.segment_po2 = segment_po2;
  lim.max_cycles = max_cycles;
  lim.keep_trace = lim.boundary_rows = trace_mode ? 1 : 0;

  // ---- the executor runs ahead on its own thread; a few finished segments wait (a 2^20-cycle segment holds 72 MiB of rows)
  std::mutex mu;
  std::condition_variable cv;
  std::deque<std::unique_ptr<Produced>> queue;
  std::vector<std::unique_ptr<Produced>> returned;  // proved segments on their way back: the executor refills their row buffers
  bool producer_done = false, stop = false;
  size_t queue_depth = 2;  // finished segments that may wait for a prover lane (set once the lanes are known)
  const char* producer_err = nullptr;
  int exit_kind = R0H_VM_LIMIT;
  uint32_t exit_code = 0;
  double executor_s = 0;
  // The row buffers are recycled, so there are a handful of them in a run; each is page-locked the first time it is seen
  // (hipHostRegister): the 72 MiB of a segment then cross PCIe by DMA at the link's rate instead of through a staging copy.  Pinning
  // 72 MiB costs tens of milliseconds and a fresh buffer as much again in page faults, so the buffers stay with the context from one
  // call to the next (RowPool): a second run starts with warm, pinned buffers.
  RowPool* pool = trace_mode ? pool_of(ctx) : nullptr;
  std::unique_lock<std::mutex> pool_lock;
  if (pool) {
    pool_lock = std::unique_lock<std::mutex>(pool->mu, std::try_to_lock);
    if (!pool_lock.owns_lock()) pool = nullptr;  // another session on this context has them: this one pins its own
  }
  if (pool) {  // (before the executor thread exists: the machine is still this thread's)  last run's buffers, if they have this run's size (the executor reserves min(2^po2, 2^22) rows)
    const size_t need = (size_t)std::min<uint64_t>((uint64_t)1 << segment_po2, (uint64_t)1 << 22);
    for (size_t k = 0; k < pool->rows.size(); k++) {
      std::vector<r0h_preflight_bound> b;
      if (k < pool->bounds.size()) b.swap(pool->bounds[k]);
      if (pool->rows[k].capacity() >= need && pool->rows[k].capacity() <= 2 * need) vm_recycle_trace(vm, pool->rows[k], b);
      else pool->unpin(pool->rows[k].data());
    }
    pool->rows.clear();
    pool->bounds.clear();
  }
  // whatever path leaves this function, after the executor thread is gone: the buffers go back to the pool, and what is pinned but
  // no longer there is unpinned before its memory is freed
  struct Collect {
    RowPool* pool; r0h_vm* vm; std::deque<std::unique_ptr<Produced>>& queue; std::vector<std::unique_ptr<Produced>>& returned;
    ~Collect() {
      if (!pool) return;
      for (auto& p : queue) if (p) { pool->rows.emplace_back(); pool->rows.back().swap(p->rows); pool->bounds.emplace_back(); pool->bounds.back().swap(p->bounds); }
      for (auto& p : returned) if (p) { pool->rows.emplace_back(); pool->rows.back().swap(p->rows); pool->bounds.emplace_back(); pool->bounds.back().swap(p->bounds); }
      vm_take_spares(vm, pool->rows, pool->bounds);
      pool->settle(6);
    }
  } collect{pool, vm, queue, returned};
  r0h_system_state first_pre;  // of segment 0: the image id (written by the executor thread, read after it is joined)
  memset(&first_pre, 0, sizeof first_pre);
  std::thread producer([&] {
    const char* err = nullptr;
    try {
      size_t handed = 0;
      for (int finished = 0; !finished && !err;) {
        std::vector<std::unique_ptr<Produced>> back;
        {
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&] { return stop || queue.size() < queue_depth; });
          if (stop) break;
          back.swap(returned);
        }
        for (auto& b : back) vm_recycle_trace(vm, b->rows, b->bounds);
        const Clock::time_point t0 = Clock::now();
        err = r0h_vm_run_segment(vm, &lim, &finished, &exit_kind, &exit_code);
        executor_s += seconds(t0, Clock::now());
        if (err) break;
        // (the end of the run may have cut more than one segment: the last cycles and the rows that close the session)
        for (; handed < r0h_vm_n_segments(vm) && !err; handed++) {
          std::unique_ptr<Produced> p(new Produced());
          const size_t i = handed;
          err = r0h_vm_segment_info(vm, i, &p->info);
          if (!err) err = r0h_vm_segment_claim(vm, i, &p->claim);
          if (err) break;
          p->index = i;
          if (i == 0) first_pre = p->info.pre;
          if (i % parts != part) {  // another rank's segment: its row buffers go straight back to the machine
            if (trace_mode) { vm_take_trace(vm, i, p->rows, p->bounds); vm_recycle_trace(vm, p->rows, p->bounds); }
            continue;
          }
          if (trace_mode) vm_take_trace(vm, i, p->rows, p->bounds);
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&] { return stop || queue.size() < queue_depth + 2; });
          if (stop) break;
          queue.push_back(std::move(p));
          cv.notify_all();
        }
      }
    } catch (const std::exception& ex) {
      err = make_error("exception in the executor: %s", ex.what());
    } catch (...) {
      err = make_error("unknown exception in the executor");
    }
    std::lock_guard<std::mutex> lk(mu);
    producer_err = err;
    producer_done = true;
    cv.notify_all();
  });
  struct Pins {
    RowPool* pool;
    std::map<const void*, size_t> local;
    std::map<const void*, size_t>& seen() { return pool ? pool->pinned : local; }
    void pin(const void* p, size_t bytes) {
      if (!p || !bytes) return;
      auto& m = seen();
      auto it = m.find(p);
      if (it != m.end() && it->second >= bytes) return;
      if (it != m.end()) { (void)hipHostUnregister(const_cast<void*>(p)); m.erase(it); }
      if (hipHostRegister(const_cast<void*>(p), bytes, hipHostRegisterDefault) == hipSuccess) m[p] = bytes;
      else (void)hipGetLastError();  // pageable memory still works, only slower
    }
    ~Pins() { for (auto& kv : local) (void)hipHostUnregister(const_cast<void*>(kv.first)); }
  } pins{pool, {}};
  struct Join {  // whatever path leaves this function: the executor thread is told to stop and joined first
    std::thread& t; std::mutex& mu; std::condition_variable& cv; bool& stop;
    ~Join() {
      { std::lock_guard<std::mutex> lk(mu); stop = true; }
      cv.notify_all();
      if (t.joinable()) t.join();
    }
  } join{producer, mu, cv, stop};

  // ---- every segment as it arrives.  Two prover lanes by default (R0H_SESSION_LANES = 1..4): lane 0 is the caller's context on the
  // calling thread, the others are helper contexts of the same device (kept with `ctx` between calls) on threads of their own -- while
  // one lane waits for a transcript read-back the other keeps the device busy.  The circuit and the CODE commitments are shared.
  uint32_t n_lanes = 2;
  if (const char* v = getenv("R0H_SESSION_LANES")) n_lanes = (uint32_t)strtoul(v, nullptr, 10);
  n_lanes = n_lanes < 1 ? 1 : n_lanes > 4 ? 4 : n_lanes;
  std::vector<r0h_ctx*>& lane_ctx = ses->lane_ctx;
  lane_ctx.assign(n_lanes, ctx);
  for (uint32_t k = 1; k < n_lanes; k++) R0H_TRY(ctx_helper(ctx, k - 1, &lane_ctx[k]));
  {
    std::lock_guard<std::mutex> lk(mu);
    queue_depth = n_lanes + 1;
    cv.notify_all();
  }
  std::mutex commit_mu, result_mu;
  CodeCommits& commits = ses->commits;
  const char* lane_err = nullptr;

  auto lane_body = [&](r0h_ctx* lctx) -> const char* {
    std::vector<uint32_t> seal, global(c->n_global);
    for (;;) {
      std::unique_ptr<Produced> seg;
      size_t i;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return !queue.empty() || producer_done || stop; });
        if (stop) return nullptr;
        if (queue.empty()) {
          if (producer_err) { const char* e = producer_err; producer_err = nullptr; return e; }
          return nullptr;
        }
        seg = std::move(queue.front());
        queue.pop_front();
        i = seg->index;
        cv.notify_all();
      }
      // whatever leaves this iteration, the segment's row buffers go back (they may be page-locked: the pool unpins what it drops)
      struct Return {
        std::unique_ptr<Produced>& seg; std::mutex& mu; std::vector<std::unique_ptr<Produced>>& returned; bool on;
        ~Return() { if (on && seg) { std::lock_guard<std::mutex> lk(mu); returned.push_back(std::move(seg)); } }
      } give_back{seg, mu, returned, trace_mode};
      const uint64_t rows_needed = trace_mode ? seg->rows.size() + seg->bounds.size() : seg->info.user_cycles + seg->info.paging_cycles;
      uint32_t po2 = trace_size(rows_needed);
      if (trace_mode && po2 < R0H_TRACE_MIN_PO2) po2 = R0H_TRACE_MIN_PO2;
      uint8_t cd[32];
      claim_digest(seg->claim, cd);
      std::fill(global.begin(), global.end(), 0u);
      claim_globals(cd, global.data());
      const size_t n = (size_t)1 << po2;
      r0h_buf* data = nullptr;
      R0H_TRY(buf_alloc_pooled(lctx, (size_t)c->group_size[R0H_GROUP_DATA] * n * 4, &data));
      struct Free { r0h_buf* b; ~Free() { if (b) r0h_buf_free(b); } } data_guard{data};
      r0h_code_commit* cc = nullptr;
      {
        std::lock_guard<std::mutex> lk(commit_mu);
        R0H_TRY(commits.get(lctx, c, po2, data, &cc));
      }
      const Clock::time_point t0 = Clock::now();
      Pending pend;
      pend.index = i;
      pend.lctx = lctx;
      pend.po2 = po2;
      pend.cc = cc;
      pend.claim = seg->claim;
      if (trace_mode) {
        {
          std::lock_guard<std::mutex> lk(result_mu);
          pins.pin(seg->rows.data(), seg->rows.capacity() * sizeof(r0h_preflight_row));
        }
        const r0h_trace_segment ts = {seg->info.index + 1, seg->info.closing, seg->info.pre.pc, 0};
        R0H_TRY(r0h_trace_witgen(lctx, seg->rows.data(), seg->rows.size(), seg->bounds.data(), seg->bounds.size(), po2, &ts, data, global.data()));
        R0H_TRY(r0h_logup_multiplicities(lctx, c, po2, data, global.data()));
        const Clock::time_point t1 = Clock::now();
        R0H_TRY(r0h_proof_begin_committed(lctx, c, po2, cc, data, global.data(), nullptr, &pend.proof));
        struct Abort { r0h_proof*& p; bool armed; ~Abort() { if (armed && p) { r0h_proof_abort(p); p = nullptr; } } } abort_guard{pend.proof, true};
        R0H_TRY(r0h_proof_data_root(pend.proof, pend.root));
        {  // beyond the session's resident limit a segment waits for the challenge without its evaluations
          const uint64_t evaluations = (uint64_t)c->group_size[R0H_GROUP_DATA] * n * 16;
          bool lean;
          {
            std::lock_guard<std::mutex> lk(result_mu);
            lean = ses->resident_evaluations + evaluations > ses->resident_limit;
            if (lean) stats.lean_segments++;
            else ses->resident_evaluations += evaluations;
          }
          if (lean) R0H_TRY(r0h_proof_shrink(pend.proof, nullptr));
        }
        const Clock::time_point t2 = Clock::now();
        pend.global = global;
        pend.data = data;
        data_guard.b = nullptr;
        abort_guard.armed = false;
        std::lock_guard<std::mutex> lk(result_mu);
        stats.witgen_ms += 1e3 * seconds(t0, t1);
        stats.prove_ms += 1e3 * seconds(t1, t2);
        ses->pending.push_back(std::move(pend));
      } else {
        // synthetic column program with the claim planted (CODE is regenerated into a scratch block: only DATA is used)
        r0h_buf* code = nullptr;
        R0H_TRY(buf_alloc_pooled(lctx, (size_t)c->group_size[R0H_GROUP_CODE] * n * 4, &code));
        const char* err = r0h_witgen_public(lctx, c, po2, 0x5E55 + i, global.data(), code, data);
        r0h_buf_free(code);
        if (err) return err;
        const Clock::time_point t1 = Clock::now();
        size_t words = 0;
        seal.resize((size_t)1 << 20);
        R0H_TRY(r0h_prove_segment_committed(lctx, c, po2, cc, data, global.data(), seal.data(), seal.size(), &words));
        const Clock::time_point t2 = Clock::now();
        pend.seal.assign(seal.begin(), seal.begin() + words);
        pend.done = true;
        std::lock_guard<std::mutex> lk(result_mu);
        stats.witgen_ms += 1e3 * seconds(t0, t1);
        stats.prove_ms += 1e3 * seconds(t1, t2);
        stats.segments++;
        ses->pending.push_back(std::move(pend));
      }
    }
  };
  auto guarded = [&](r0h_ctx* lctx) {
    const char* err = nullptr;
    try {
      err = lane_body(lctx);
    } catch (const std::exception& ex) {
      err = make_error("exception in a prover lane: %s", ex.what());
    } catch (...) {
      err = make_error("unknown exception in a prover lane");
    }
    if (err) {
      std::lock_guard<std::mutex> lk(mu);
      if (!lane_err) lane_err = err;
      else r0h_free_error(err);
      stop = true;  // the other lanes and the executor leave at their next look
      cv.notify_all();
    }
  };
  {
    std::vector<std::thread> workers;
    for (uint32_t k = 1; k < n_lanes; k++) workers.emplace_back(guarded, lane_ctx[k]);
    guarded(lane_ctx[0]);
    for (std::thread& t : workers) t.join();
  }
  if (lane_err) return lane_err;
  producer.join();  // finished: the machine is this thread's again
  R0H_REQUIRE(exit_kind != R0H_VM_LIMIT, "r0h_prove_elf: the guest did not halt within %llu cycles (session limit)", (unsigned long long)max_cycles);
  R0H_REQUIRE(exit_code == 0, "r0h_prove_elf: the guest exited with code %u", exit_code);  // `prove` is an Err for a failed guest
  ses->n_segments = r0h_vm_n_segments(vm);
  std::sort(ses->pending.begin(), ses->pending.end(), [](const Pending& a, const Pending& b) { return a.index < b.index; });
  size_t own = 0;
  for (size_t i = part; i < ses->n_segments; i += parts) own++;
  R0H_REQUIRE(ses->pending.size() == own, "r0h_prove_elf: %zu of this rank's %zu segments were committed", ses->pending.size(), own);
  ses->cycles = r0h_vm_cycles(vm);
  const uint8_t* journal; size_t journal_len;
  R0H_TRY(r0h_vm_journal(vm, &journal, &journal_len));
  ses->journal.assign(journal, journal + journal_len);
  system_state_digest(first_pre, ses->image_id);  // the image id the verifier is given: digest of the state the run started from
  stats.cycles = ses->cycles;
  stats.executor_s = executor_s;
  *session_out = ses.release();
  return nullptr;
  R0H_GUARD_END
}

size_t r0h_session_n_segments(const r0h_session* s) { return s ? s->n_segments : 0; }

// what this rank's segments contribute to the session challenge: R0H_SESSION_RECORD_WORDS words each (early public inputs, DATA root),
// with their indices; a circuit without a session argument contributes nothing
const char* r0h_session_records(const r0h_session* s, uint32_t* indices_out, uint32_t* records_out, size_t capacity, size_t* n_out) {
  R0H_REQUIRE(s && n_out, "r0h_session_records: NULL argument");
  *n_out = s->trace_mode ? s->pending.size() : 0;
  if (!s->trace_mode) return nullptr;
  R0H_REQUIRE(capacity >= s->pending.size() && indices_out && records_out, "r0h_session_records: room for %zu records wanted", s->pending.size());
  const uint32_t n_early = R0H_TRACE_GLOBALS - R0H_TRACE_LATE_GLOBALS;
  for (size_t k = 0; k < s->pending.size(); k++) {
    indices_out[k] = (uint32_t)s->pending[k].index;
    memcpy(records_out + k * RECORD_WORDS, s->pending[k].global.data(), n_early * 4);
    memcpy(records_out + k * RECORD_WORDS + n_early, s->pending[k].root, 32);
  }
  return nullptr;
}

// Phase 2: with the records of ALL segments of the session (in index order) the challenge is fixed; every proof of this rank receives
// it and its own sum under it as late public inputs and is finished.  The receipt holds this rank's segments.
const char* r0h_session_finish(r0h_session* s, const uint32_t* all_records, size_t n_records, r0h_receipt** receipt_out, uint8_t image_id_out[32], uint64_t* cycles_out) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(s && receipt_out, "r0h_session_finish: NULL argument");
  const r0h_circuit* c = s->c;
  std::vector<uint32_t> image_seal;
  if (s->trace_mode) {
    R0H_REQUIRE(all_records && n_records == s->n_segments, "r0h_session_finish: the session has %zu segments, %zu records were given", s->n_segments, n_records);
    const uint32_t n_early = R0H_TRACE_GLOBALS - R0H_TRACE_LATE_GLOBALS;
    for (const Pending& p : s->pending)
      R0H_REQUIRE(!memcmp(all_records + p.index * RECORD_WORDS, p.global.data(), n_early * 4) && !memcmp(all_records + p.index * RECORD_WORDS + n_early, p.root, 32),
                  "r0h_session_finish: record %zu is not the one this rank committed", p.index);
    uint32_t challenge[16];
    session_challenge(all_records, n_records, challenge);
    std::mutex err_mu;
    const char* first_err = nullptr;
    auto lane = [&](r0h_ctx* lctx) {
      std::vector<uint32_t> seal((size_t)1 << 20), mix(c->n_mix);
      if (lctx == s->ctx && !s->elf.empty() && s->ctx->image_circuit) {
        // the image's side of the balance, proved (`receipt.verify(image_id)` then needs no ELF): first thing on the first lane, beside
        // the other lanes' segments -- all it waits for is the challenge
        size_t words = 0;
        const char* err = r0h_prove_image(s->ctx, s->ctx->image_circuit, s->elf.data(), s->elf.size(), challenge, seal.data(), seal.size(), &words);
        std::lock_guard<std::mutex> lk(err_mu);
        if (err) { if (!first_err) first_err = err; else r0h_free_error(err); return; }
        image_seal.assign(seal.begin(), seal.begin() + words);
      }
      for (Pending& p : s->pending) {
        if (p.lctx != lctx || p.done) continue;
        {
          std::lock_guard<std::mutex> lk(err_mu);
          if (first_err) return;
        }
        const Clock::time_point t0 = Clock::now();
        const char* err = nullptr;
        const size_t n = (size_t)1 << p.po2;
        r0h_buf* accum = nullptr;
        size_t words = 0;
        do {
          memcpy(p.global.data() + R0H_TRACE_GAMMA, challenge, 64);
          const r0h_buf* code_cols = nullptr;
          if ((err = r0h_code_commit_columns(p.cc, &code_cols))) break;
          if ((err = r0h_logup_totals(lctx, c, p.po2, code_cols, p.data, p.global.data()))) break;
          if ((err = r0h_proof_late(p.proof, p.global.data() + (R0H_TRACE_GLOBALS - R0H_TRACE_LATE_GLOBALS), mix.data()))) break;
          if ((err = buf_alloc_pooled(lctx, (size_t)c->group_size[R0H_GROUP_ACCUM] * n * 4, &accum))) break;
          if ((err = r0h_accum_public(lctx, c, p.po2, code_cols, p.data, p.global.data(), mix.data(), accum))) break;
          r0h_proof* proof = p.proof;
          p.proof = nullptr;  // consumed either way
          if ((err = r0h_proof_finish(proof, accum, seal.data(), seal.size(), &words))) break;
        } while (false);
        if (accum) r0h_buf_free(accum);
        r0h_buf_free(p.data);
        p.data = nullptr;
        std::lock_guard<std::mutex> lk(err_mu);
        if (err) { if (!first_err) first_err = err; else r0h_free_error(err); return; }
        p.seal.assign(seal.begin(), seal.begin() + words);
        p.done = true;
        s->stats.prove_ms += 1e3 * seconds(t0, Clock::now());
        s->stats.segments++;
      }
    };
    auto guarded = [&](r0h_ctx* lctx) {
      try {
        lane(lctx);
      } catch (const std::exception& ex) {
        std::lock_guard<std::mutex> lk(err_mu);
        if (!first_err) first_err = make_error("exception in a prover lane: %s", ex.what());
      }
    };
    std::vector<std::thread> workers;
    for (size_t k = 1; k < s->lane_ctx.size(); k++) workers.emplace_back(guarded, s->lane_ctx[k]);
    guarded(s->lane_ctx[0]);
    for (std::thread& t : workers) t.join();
    if (first_err) return first_err;
  }
  r0h_receipt* rc = nullptr;
  R0H_TRY(r0h_receipt_new(R0H_RECEIPT_COMPOSITE, nullptr, 0, &rc));
  std::unique_ptr<r0h_receipt, const char* (*)(r0h_receipt*)> rc_guard(rc, r0h_receipt_free);
  for (Pending& p : s->pending) {
    R0H_REQUIRE(p.done, "r0h_prove_elf: segment %zu was never proved", p.index);
    R0H_TRY(r0h_receipt_add_segment_claim(rc, p.seal.data(), p.seal.size(), (uint32_t)p.index, &p.claim, nullptr));
  }
  rc->journal = s->journal;
  rc->image_seal = image_seal;
  if (image_id_out) memcpy(image_id_out, s->image_id, 32);
  if (cycles_out) *cycles_out = s->cycles;
  s->stats.wall_s = seconds(s->t_begin, Clock::now());
  s->ctx->session = s->stats;
  *receipt_out = rc_guard.release();
  return nullptr;
  R0H_GUARD_END
}

const char* r0h_prove_elf(r0h_ctx* ctx, const r0h_circuit* c, const uint8_t* elf, size_t elf_len, const uint32_t* input_words, size_t n_input, uint32_t segment_po2,
                          uint64_t max_cycles, r0h_receipt** receipt_out, uint8_t image_id_out[32], uint64_t* cycles_out) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(receipt_out, "r0h_prove_elf: NULL argument");
  r0h_session* s = nullptr;
  R0H_TRY(r0h_session_begin(ctx, c, elf, elf_len, input_words, n_input, segment_po2, max_cycles, 0, 1, &s));
  std::unique_ptr<r0h_session> guard(s);
  std::vector<uint32_t> indices(s->pending.size()), records(s->pending.size() * RECORD_WORDS);
  size_t n = 0;
  R0H_TRY(r0h_session_records(s, indices.data(), records.data(), s->pending.size(), &n));
  return r0h_session_finish(s, records.data(), n, receipt_out, image_id_out, cycles_out);
  R0H_GUARD_END
}

// One rank's share of a session whose circuit has no session-wide argument (the synthetic circuits: segments are independent);
// with the trace circuit the ranks exchange their records between the phases: r0h_session_begin / _records / _finish.
const char* r0h_prove_elf_part(r0h_ctx* ctx, const r0h_circuit* c, const uint8_t* elf, size_t elf_len, const uint32_t* input_words, size_t n_input, uint32_t segment_po2,
                               uint64_t max_cycles, uint32_t part, uint32_t parts, r0h_receipt** receipt_out, uint8_t image_id_out[32], uint64_t* cycles_out) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(receipt_out && c, "r0h_prove_elf_part: NULL argument");
  R0H_REQUIRE(parts == 1 || c->n_late == 0, "r0h_prove_elf_part: the segments of a trace-circuit session share one challenge: ranks use r0h_session_begin / r0h_session_records / r0h_session_finish");
  r0h_session* s = nullptr;
  R0H_TRY(r0h_session_begin(ctx, c, elf, elf_len, input_words, n_input, segment_po2, max_cycles, part, parts, &s));
  std::unique_ptr<r0h_session> guard(s);
  std::vector<uint32_t> indices(s->pending.size() + 1), records((s->pending.size() + 1) * RECORD_WORDS);
  size_t n = 0;
  R0H_TRY(r0h_session_records(s, indices.data(), records.data(), s->pending.size(), &n));
  return r0h_session_finish(s, records.data(), n, receipt_out, image_id_out, cycles_out);
  R0H_GUARD_END
}

}  // extern "C"
