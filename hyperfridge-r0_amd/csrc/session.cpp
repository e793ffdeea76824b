// `Prover::prove(env, elf)` in the shape hyperfridge calls it (host/src/main.rs:420-423): execute the guest, cut the run
// into segments, prove every segment, assemble the composite receipt.  risc0-zkvm 3.0.5 `LocalProver::prove` = executor
// (`ExecutorImpl::run` -> Session{segments}) + `prove_session` (one `prove_segment` per segment) + CompositeReceipt.
//
// What each stage is here: the executor is csrc/rv32im.cpp (RV32IM by the specification; this library's ecall ABI and cycle model,
// see there); `prove_segment` is the device-resident sequencer (csrc/prover.hip) at the trace size the segment needs; the claim of
// every segment (system states from the run, exit code, the journal's output digest on the last one) is bound to its seal through
// the public inputs (csrc/claim.cpp).
//
// With the trace circuit (circuits/trace.r0c, ProtocolInfo "R0HIP_TRACE:v3__") the seal of a segment attests THAT segment: the
// executor keeps one compact row per cycle plus one per address touched, the device expands them into the DATA group
// (csrc/trace.hip: 72 bytes per cycle cross PCIe, not the 576 bytes of an expanded row), and the proof is over those columns --
// contiguity, control flow and memory consistency as include/r0hip.h lists them (what an instruction computes is risc0's rv32im
// circuit and stays unconstrained).  The guest runs ahead on its own host thread; two prover lanes (the caller's context and a
// helper context of the same device) take the segments as they are cut.  With any other circuit the witness is the blob's synthetic column program with the claim
// planted: the seal then proves "a satisfying trace of the loaded circuit exists whose public inputs name this claim", not
// "this program ran".
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

extern "C" {

const char* r0h_last_session_stats(r0h_ctx* ctx, r0h_session_stats* out) {
  R0H_REQUIRE(ctx && out, "r0h_last_session_stats: NULL argument");
  *out = ctx->session;
  return nullptr;
}

const char* r0h_prove_elf(r0h_ctx* ctx, const r0h_circuit* c, const uint8_t* elf, size_t elf_len, const uint32_t* input_words, size_t n_input, uint32_t segment_po2,
                          uint64_t max_cycles, r0h_receipt** receipt_out, uint8_t image_id_out[32], uint64_t* cycles_out) {
  return r0h_prove_elf_part(ctx, c, elf, elf_len, input_words, n_input, segment_po2, max_cycles, 0, 1, receipt_out, image_id_out, cycles_out);
}

// One rank's share of a session: the guest is executed in full (it is deterministic and takes a tenth of a second per ten million
// cycles -- cheaper than shipping 72 MiB of rows per segment to another GPU), segments part, part + parts, ... are proved; the
// receipt carries those segments only (r0h_receipt_merge puts the ranks' receipts together).
const char* r0h_prove_elf_part(r0h_ctx* ctx, const r0h_circuit* c, const uint8_t* elf, size_t elf_len, const uint32_t* input_words, size_t n_input, uint32_t segment_po2,
                               uint64_t max_cycles, uint32_t part, uint32_t parts, r0h_receipt** receipt_out, uint8_t image_id_out[32], uint64_t* cycles_out) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(ctx && c && elf && receipt_out && (input_words || !n_input), "r0h_prove_elf: NULL argument");
  R0H_REQUIRE(parts >= 1 && part < parts, "r0h_prove_elf_part: part %u of %u", part, parts);
  const bool trace_mode = !memcmp(c->info, "R0HIP_TRACE:v3__", 16);
  R0H_REQUIRE(segment_po2 >= 9 && segment_po2 <= (trace_mode ? R0H_TRACE_MAX_PO2 : R0H_MAX_PO2), "r0h_prove_elf: segment_po2 %u outside [9, %u]", segment_po2,
              (unsigned)(trace_mode ? R0H_TRACE_MAX_PO2 : R0H_MAX_PO2));
  R0H_REQUIRE(c->has_column_program, "r0h_prove_elf: the circuit has no column program (CODE columns, accumulation)");
  R0H_REQUIRE(c->n_global >= 8, "r0h_prove_elf: the circuit exposes %u public inputs, a claim needs 8", c->n_global);
  if (trace_mode) R0H_REQUIRE(c->n_global == R0H_TRACE_GLOBALS && c->group_size[R0H_GROUP_DATA] == R0H_TRACE_COLUMNS, "r0h_prove_elf: this trace circuit is not the one the library was built for");
  if (!max_cycles) max_cycles = R0H_DEFAULT_SESSION_LIMIT;  // a guest that never halts must not hang the host (risc0: session limit)
  const Clock::time_point t_begin = Clock::now();
  r0h_session_stats stats = {0, 0, 0, 0, 0, 0};
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

  // ---- the executor runs ahead on its own thread; at most two finished segments wait (a 2^20-cycle segment holds 72 MiB of rows)
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
  // The row buffers are recycled, so there are three or four of them in a run; each is page-locked the first time it is seen
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
        std::unique_ptr<Produced> p(new Produced());
        const size_t i = r0h_vm_n_segments(vm) - 1;
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
        std::lock_guard<std::mutex> lk(mu);
        queue.push_back(std::move(p));
        cv.notify_all();
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

  // ---- prove every segment for its claim, as it arrives.  Two prover lanes by default (R0H_SESSION_LANES = 1..4): lane 0 is the caller's
  // context on the calling thread, the others are helper contexts of the same device (kept with `ctx` between calls) on threads of
  // their own -- while one lane waits for a transcript read-back the other keeps the device busy.  The circuit and the CODE
  // commitments are shared; a segment's seal lands at its index whichever lane made it.
  uint32_t n_lanes = 2;
  if (const char* v = getenv("R0H_SESSION_LANES")) n_lanes = (uint32_t)strtoul(v, nullptr, 10);
  n_lanes = n_lanes < 1 ? 1 : n_lanes > 4 ? 4 : n_lanes;
  std::vector<r0h_ctx*> lane_ctx(n_lanes, ctx);
  for (uint32_t k = 1; k < n_lanes; k++) R0H_TRY(ctx_helper(ctx, k - 1, &lane_ctx[k]));
  {
    std::lock_guard<std::mutex> lk(mu);
    queue_depth = n_lanes + 1;
    cv.notify_all();
  }
  struct Proved { std::vector<uint32_t> seal; r0h_receipt_claim claim; bool done = false; };
  std::vector<Proved> proved;
  std::mutex commit_mu, result_mu;
  CodeCommits commits;
  const char* lane_err = nullptr;

  auto lane_body = [&](r0h_ctx* lctx) -> const char* {
    std::vector<uint32_t> seal((size_t)1 << 20), global(c->n_global);
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
      const uint64_t rows_needed = trace_mode ? seg->rows.size() + seg->bounds.size() : seg->info.user_cycles + seg->info.paging_cycles;
      const uint32_t po2 = trace_size(rows_needed);
      uint8_t cd[32];
      claim_digest(seg->claim, cd);
      std::fill(global.begin(), global.end(), 0u);
      claim_globals(cd, global.data());
      const size_t n = (size_t)1 << po2;
      r0h_buf* data = nullptr;
      R0H_TRY(buf_alloc_pooled(lctx, (size_t)c->group_size[R0H_GROUP_DATA] * n * 4, &data));
      struct Free { r0h_buf* b; ~Free() { r0h_buf_free(b); } } data_guard{data};
      r0h_code_commit* cc = nullptr;
      {
        std::lock_guard<std::mutex> lk(commit_mu);
        R0H_TRY(commits.get(lctx, c, po2, data, &cc));
      }
      const Clock::time_point t0 = Clock::now();
      size_t words = 0;
      if (trace_mode) {
        {
          std::lock_guard<std::mutex> lk(result_mu);
          pins.pin(seg->rows.data(), seg->rows.capacity() * sizeof(r0h_preflight_row));
        }
        R0H_TRY(r0h_trace_witgen(lctx, seg->rows.data(), seg->rows.size(), seg->bounds.data(), seg->bounds.size(), po2, data, global.data()));
      } else {
        // synthetic column program with the claim planted (CODE is regenerated into a scratch block: only DATA is used)
        r0h_buf* code = nullptr;
        R0H_TRY(buf_alloc_pooled(lctx, (size_t)c->group_size[R0H_GROUP_CODE] * n * 4, &code));
        const char* err = r0h_witgen_public(lctx, c, po2, 0x5E55 + i, global.data(), code, data);
        r0h_buf_free(code);
        if (err) return err;
      }
      const Clock::time_point t1 = Clock::now();
      R0H_TRY(r0h_prove_segment_committed(lctx, c, po2, cc, data, global.data(), seal.data(), seal.size(), &words));
      const Clock::time_point t2 = Clock::now();
      {
        std::lock_guard<std::mutex> lk(result_mu);
        stats.witgen_ms += 1e3 * seconds(t0, t1);
        stats.prove_ms += 1e3 * seconds(t1, t2);
        stats.segments++;
        if (proved.size() <= i) proved.resize(i + 1);
        proved[i].seal.assign(seal.begin(), seal.begin() + words);
        proved[i].claim = seg->claim;
        proved[i].done = true;
      }
      if (trace_mode) {  // the row buffers go back to the executor
        std::lock_guard<std::mutex> lk(mu);
        returned.push_back(std::move(seg));
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
  r0h_receipt* rc = nullptr;
  R0H_TRY(r0h_receipt_new(R0H_RECEIPT_COMPOSITE, nullptr, 0, &rc));
  std::unique_ptr<r0h_receipt, const char* (*)(r0h_receipt*)> rc_guard(rc, r0h_receipt_free);
  producer.join();  // finished: the machine is this thread's again
  const size_t n_segments = r0h_vm_n_segments(vm);
  for (size_t i = part; i < n_segments; i += parts) {
    R0H_REQUIRE(i < proved.size() && proved[i].done, "r0h_prove_elf: segment %zu was never proved", i);
    R0H_TRY(r0h_receipt_add_segment_claim(rc, proved[i].seal.data(), proved[i].seal.size(), (uint32_t)i, &proved[i].claim, nullptr));
  }
  R0H_REQUIRE(exit_kind != R0H_VM_LIMIT, "r0h_prove_elf: the guest did not halt within %llu cycles (session limit)", (unsigned long long)max_cycles);
  R0H_REQUIRE(exit_code == 0, "r0h_prove_elf: the guest exited with code %u", exit_code);  // `prove` is an Err for a failed guest
  if (cycles_out) *cycles_out = r0h_vm_cycles(vm);
  const uint8_t* journal; size_t journal_len;
  R0H_TRY(r0h_vm_journal(vm, &journal, &journal_len));
  rc->journal.assign(journal, journal + journal_len);
  // the image id the verifier is given: digest of the state the run started from
  if (image_id_out) system_state_digest(first_pre, image_id_out);
  stats.cycles = r0h_vm_cycles(vm);
  stats.executor_s = executor_s;
  stats.wall_s = seconds(t_begin, Clock::now());
  ctx->session = stats;
  *receipt_out = rc_guard.release();
  return nullptr;
  R0H_GUARD_END
}

}  // extern "C"
