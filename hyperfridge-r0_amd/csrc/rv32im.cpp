// RV32IM executor, segmenter and preflight trace (host-only, no device work) -- SURVEY.md 8(f) rank 2, the groundwork for the row
// "executor + preflight": the part of `prover.prove(env, HYPERFRIDGE_ELF)` (host/src/main.rs:423) that runs BEFORE prove_segment --
// risc0-circuit-rv32im 4.0.4's emulator executes the guest ELF, cuts the run into segments of at most 2^po2 cycles
// (`segment_count = 37` for this guest at risc0 0.19: docs/runtime.md:48-51) and records, per segment, the trace the witness
// generator replays.
//
// What is pinned and what is not.  The instruction semantics are the RISC-V unprivileged ISA (RV32I + M), a public specification:
// the tests run hand-encoded instruction words and compare with an independent Python interpreter, including the M-extension corner
// cases the specification tabulates (division by zero, signed overflow).  Everything risc0-specific is RECALLED in outline only and
// written here as this library's own, documented choice -- a real guest ELF needs risc0's tables instead:
//   * the ecall ABI (below) is NOT risc0's syscall table (halt / software / sha / bigint selected by t0);
//   * the cycle model is one cycle per instruction -- one per WORD moved for the two I/O ecalls, which re-execute like `rep movs`
//     until their count is zero, so that a cycle has at most one memory access -- plus, per segment, a flat charge per page first
//     touched / dirtied (page_in_cycles / page_out_cycles) and, with boundary_rows, one row per distinct register or memory word
//     touched (what the trace circuit spends on each address's first and last value); risc0's rv32im-v2 charges differ and are
//     not reproducible from the reference;
//   * the state digest is risc0-binfmt's SystemState{pc, merkle_root} (csrc/claim.cpp) with merkle_root = a SHA-256 binary Merkle
//     tree over the 1 KiB pages of the 32-bit address space, all-zero subtrees folded (risc0's image id is also a page Merkle root;
//     its exact tree shape and tags are not pinned here).
// No guest ELF exists in the reference (only sources: methods/guest/src/main.rs; the ELF is built by `risc0_build::embed_methods()`,
// methods/build.rs:2, which needs the Rust toolchain), so the camt53 trace itself still cannot be produced.
//
// ecall ABI (a7 = x17 selects; arguments a0, a1; every ecall cycle reads a7 and a0 -- they are the cycle's two register reads):
//   0 HALT        a0 = exit code                        -- ends the run (ExitCode::Halted(a0))
//   1 READ_WORDS  a0 = destination, a1 = word count     -- the next words of the input stream (ExecutorEnv frames), zero past its end
//   2 COMMIT      a0 = source, a1 = word count          -- journal words (`env::commit`).  The journal is a WINDOW of guest memory: the
//                                                          word at R0H_JOURNAL_BASE + 4 i is journal word i, committed once; a run's
//                                                          commits together cover [0, n) -- so that a verifier who knows only the
//                                                          journal knows which (address, word) pairs the COMMIT rows name (csrc/claim.cpp)
//   3 CYCLES                                            -- a0 = cycles executed so far (`env::cycle_count()`)
//   4 PAUSE       a0 = exit code                        -- ends the run resumably (ExitCode::Paused(a0))
// The two transfers move one word per cycle and keep no state outside the registers, so that a cycle is a function of what it
// reads (the trace circuit constrains it): while a1 = j > 0 the ecall re-executes -- it moves word j - 1 of the buffer (address
// a0 + 4 (j - 1)) and writes a1 = j - 1 -- and with a1 = 0 it falls through to pc + 4; a transfer of n words takes n + 1 cycles.
// The host hands out the stream words so that the buffer ends up in stream order.  a0 is word-aligned, buffers lie below 1 GiB.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <array>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/r0hip.h"
#include "internal.hpp"
#include "receipt_types.hpp"
#include "trace.hpp"

namespace {
// Guest memory is the low 1 GiB: the trace circuit carries a pc as one field element and keeps registers above 2^28 word addresses
// (R0H_REG_BASE); an access above it is a guest trap like a misaligned one.
constexpr uint32_t ADDRESS_BITS = 30;
static_assert(R0H_REG_BASE == 1u << (ADDRESS_BITS - 2), "registers sit right above the memory words");
constexpr uint32_t PAGE_BYTES = 1024, PAGE_WORDS = PAGE_BYTES / 4, PAGE_SHIFT = 10, N_PAGE_BITS = 32 - PAGE_SHIFT;  // 2^22 pages
constexpr uint64_t MAX_JOURNAL_BYTES = (uint64_t)1 << 28, MAX_IO_WORDS = (uint64_t)1 << 26;
constexpr size_t MAX_RESIDENT_PAGES = (size_t)1 << 21;  // 2 GiB of guest memory: a run that wants more is refused, not swapped

struct Page {
  uint32_t w[PAGE_WORDS];
  uint32_t ts[PAGE_WORDS];  // timestamp of the last access in segment `epoch` (0 = not touched yet)
  uint32_t seg[PAGE_WORDS]; // number (index + 1) of the last segment that touched the word, 0 = none yet: what a boundary row names
  uint32_t epoch, in_epoch, out_epoch;  // segment (index + 1) for which ts / "paged in" / "dirtied" hold
  uint32_t stale;           // 1: written since its leaf of the memory tree was last hashed (it is then on the vm's stale list)
};

struct Segment {
  r0h_vm_segment info;
  std::vector<r0h_preflight_row> rows;
  std::vector<r0h_preflight_bound> bounds;
};
struct Run;
}  // namespace

struct r0h_vm {
  uint32_t x[32] = {0};
  uint32_t pc = 0;
  Page** table = nullptr;            // page index -> page; absent = all zero (a flat 2^22-entry table, lazily backed)
  std::vector<uint32_t> page_list;   // the pages that exist, in creation order
  // the memory Merkle tree, kept between segment boundaries: nodes of non-zero subtrees by (level << 32 | index); a boundary
  // re-hashes only the pages written since the previous one and the paths above them (risc0 keeps its page table the same way)
  std::unordered_map<uint64_t, std::array<uint8_t, 32>> tree;
  std::vector<uint32_t> stale_pages;
  void mark_stale(Page* p, uint32_t idx) {
    if (!p->stale) { p->stale = 1; stale_pages.push_back(idx); }
  }
  std::vector<uint32_t> input;
  size_t input_pos = 0;
  std::vector<uint8_t> journal;
  std::vector<uint8_t> journal_seen;  // per journal word: committed already
  uint32_t reg_seg[32] = {0};         // per register: the last segment that touched it (0 = none yet)
  std::vector<std::pair<uint32_t, uint32_t>> image;  // (word index, word) of everything loaded before the run: the program image
  bool image_sorted = false;
  uint64_t cycles = 0;  // cycles (rows) over the whole run
  bool finished = false; // a run ended in HALT / PAUSE / the cycle limit: its segments are final
  std::vector<Segment> segments;
  std::unique_ptr<Run> run;  // the run in progress (r0h_vm_run_segment)
  // row buffers handed back by whoever took a segment's trace (vm_recycle_trace): the next segment writes into memory that is
  // already mapped -- a fresh 72 MiB block costs about as much in page faults as the cycles that fill it
  std::vector<std::vector<r0h_preflight_row>> spare_rows;
  std::vector<std::vector<r0h_preflight_bound>> spare_bounds;
  // an I/O ecall in progress: the count it started with (a1 counts down from there)
  bool io_active = false;  // host-side bookkeeping of a transfer in progress: its length and where its words go in the stream
  uint32_t io_total = 0;
  size_t io_base = 0;
  // hashing caches
  uint8_t zero_level[N_PAGE_BITS + 1][32];
  bool zero_ready = false;

  r0h_vm() { table = (Page**)calloc((size_t)1 << N_PAGE_BITS, sizeof(Page*)); }
  ~r0h_vm();
  r0h_vm(const r0h_vm&) = delete;
  Page* find(uint32_t idx) const { return table ? table[idx] : nullptr; }
  Page* get(uint32_t idx) {
    Page* p = table[idx];
    if (!p) {
      if (page_list.size() >= MAX_RESIDENT_PAGES) throw std::runtime_error("the guest's memory exceeds 2 GiB of resident pages");
      p = (Page*)calloc(1, sizeof(Page));
      if (!p) throw std::bad_alloc();
      table[idx] = p;
      page_list.push_back(idx);
      // a new page is all zero, as the tree already assumes: nothing to re-hash until it is written
    }
    return p;
  }
};

namespace {
using r0h::sha256;

void page_hash(const uint32_t* w, uint8_t out[32]) { sha256(w, PAGE_BYTES, out); }  // little-endian words = the bytes of memory
void node_hash(const uint8_t a[32], const uint8_t b[32], uint8_t out[32]) {
  uint8_t cat[64];
  memcpy(cat, a, 32);
  memcpy(cat + 32, b, 32);
  sha256(cat, 64, out);
}
void ensure_zero_levels(r0h_vm& vm) {
  if (vm.zero_ready) return;
  const std::vector<uint32_t> z(PAGE_WORDS, 0);
  page_hash(z.data(), vm.zero_level[0]);
  for (uint32_t l = 1; l <= N_PAGE_BITS; l++) node_hash(vm.zero_level[l - 1], vm.zero_level[l - 1], vm.zero_level[l]);
  vm.zero_ready = true;
}
// Merkle root over all 2^22 pages, all-zero subtrees taken from the table of zero levels.  Incremental: only the pages written
// since the last call are hashed again, then the 22 nodes above each (shared parents once): cost ~ (pages written) x (17 + 2 x 22)
// SHA-256 blocks per segment boundary instead of a pass over the whole image.
void memory_root(r0h_vm& vm, uint8_t out[32]) {
  ensure_zero_levels(vm);
  auto key = [](uint32_t level, uint32_t idx) { return (uint64_t)level << 32 | idx; };
  std::vector<uint32_t> level(vm.stale_pages), up;
  vm.stale_pages.clear();
  std::sort(level.begin(), level.end());
  for (uint32_t idx : level) {
    Page* p = vm.table[idx];
    p->stale = 0;
    bool nz = false;
    for (uint32_t w : p->w) nz |= w != 0;
    if (!nz) { vm.tree.erase(key(0, idx)); continue; }
    page_hash(p->w, vm.tree[key(0, idx)].data());
  }
  for (uint32_t l = 0; l < N_PAGE_BITS; l++) {
    up.clear();
    for (size_t i = 0; i < level.size(); i++) {
      const uint32_t parent = level[i] >> 1;
      if (!up.empty() && up.back() == parent) continue;
      up.push_back(parent);
      auto a = vm.tree.find(key(l, parent << 1)), b = vm.tree.find(key(l, parent << 1 | 1));
      if (a == vm.tree.end() && b == vm.tree.end()) { vm.tree.erase(key(l + 1, parent)); continue; }
      const uint8_t* left = a == vm.tree.end() ? vm.zero_level[l] : a->second.data();
      const uint8_t* right = b == vm.tree.end() ? vm.zero_level[l] : b->second.data();
      std::array<uint8_t, 32> h;
      node_hash(left, right, h.data());
      vm.tree[key(l + 1, parent)] = h;
    }
    level.swap(up);
  }
  auto top = vm.tree.find(key(N_PAGE_BITS, 0));
  memcpy(out, top == vm.tree.end() ? vm.zero_level[N_PAGE_BITS] : top->second.data(), 32);
}

// the program image by address; later loads of a word replace earlier ones
void sort_image(r0h_vm& vm) {
  if (vm.image_sorted) return;
  std::stable_sort(vm.image.begin(), vm.image.end(), [](const std::pair<uint32_t, uint32_t>& a, const std::pair<uint32_t, uint32_t>& b) { return a.first < b.first; });
  std::vector<std::pair<uint32_t, uint32_t>> uniq;
  for (const auto& e : vm.image) {
    if (!uniq.empty() && uniq.back().first == e.first) uniq.back() = e;
    else uniq.push_back(e);
  }
  vm.image.swap(uniq);
  vm.image_sorted = true;
}

// The root of the memory a run starts from -- what the image id names -- is not the page tree's: it is the Poseidon2 digest of the
// image as the list of its words in address order (tools/image_circuit.py: blocks of four (word index, low half, high half) and a
// mask), which the image circuit ties to the image's side of the session's memory argument.  A machine that has already run
// (resumed after a PAUSE) starts from the page tree's root like any later segment.
void initial_root(r0h_vm& vm, uint8_t out[32]) {
  if (vm.cycles) { memory_root(vm, out); return; }
  sort_image(vm);
  uint32_t digest[8];
  r0h::image_digest(vm.image, digest);
  for (int i = 0; i < 8; i++) {
    const uint32_t w = r0h::dec(digest[i]);
    for (int b = 0; b < 4; b++) out[4 * i + b] = (uint8_t)(w >> (8 * b));
  }
}

struct Touched { uint32_t addr, first_value, prev_seg; };
inline bool operator<(const Touched& a, const Touched& b) { return a.addr < b.addr; }

struct Run {
  r0h_vm& vm;
  const r0h_vm_limits lim;
  Segment cur;
  uint32_t epoch = 0;                  // index of the current segment + 1
  uint32_t n_in = 0, n_out = 0;        // pages first touched / dirtied in the current segment
  uint32_t reg_ts[32], reg_first[32], reg_prev_seg[32], reg_mask = 0;
  std::vector<Touched> touched;        // every word first touched in the segment: what was found there, which segment left it
  uint64_t seg_budget, executed = 0;
  const char* err = nullptr;
  r0h_preflight_row* row = nullptr;
  int exit_kind = R0H_VM_LIMIT;
  uint32_t exit_code = 0;
  uint8_t root_now[32];                // memory root at the last segment boundary (the next segment starts from it)
  bool have_root = false;

  Run(r0h_vm& v, const r0h_vm_limits& l) : vm(v), lim(l), seg_budget((uint64_t)1 << l.segment_po2) { sort_image(vm); }

  uint64_t paging_cycles() const { return (uint64_t)n_in * lim.page_in_cycles + (uint64_t)n_out * lim.page_out_cycles; }
  uint64_t boundary_count() const { return lim.boundary_rows ? touched.size() + (uint64_t)__builtin_popcount(reg_mask) : 0; }

  void begin_segment() {
    cur = Segment();
    memset(&cur.info, 0, sizeof cur.info);
    cur.info.index = (uint32_t)vm.segments.size();
    cur.info.pre.pc = vm.pc;
    if (!have_root) { initial_root(vm, root_now); have_root = true; }
    memcpy(cur.info.pre.merkle_root, root_now, 32);
    epoch = cur.info.index + 1;
    n_in = n_out = 0;
    reg_mask = 0;
    memset(reg_ts, 0, sizeof reg_ts);
    touched.clear();
    if (lim.keep_trace) {
      if (!vm.spare_rows.empty()) { cur.rows.swap(vm.spare_rows.back()); vm.spare_rows.pop_back(); cur.rows.clear(); }
      if (!vm.spare_bounds.empty()) { cur.bounds.swap(vm.spare_bounds.back()); vm.spare_bounds.pop_back(); cur.bounds.clear(); }
      cur.rows.reserve((size_t)std::min<uint64_t>(seg_budget, (uint64_t)1 << 22));
    }
  }
  // one boundary row per address the segment touched, in increasing address order (registers sit above memory)
  void own_bounds(std::vector<r0h_preflight_bound>& out) {
    std::sort(touched.begin(), touched.end());
    out.reserve(touched.size() + 32);
    for (const Touched& t : touched) {
      const Page* p = vm.table[t.addr >> 8];
      out.push_back(r0h_preflight_bound{t.addr, t.first_value, p->w[t.addr & 255], p->ts[t.addr & 255], t.prev_seg, 0, 0, 0});
    }
    for (uint32_t i = 0; i < 32; i++)
      if (reg_mask >> i & 1) out.push_back(r0h_preflight_bound{R0H_REG_BASE + i, reg_first[i], vm.x[i], reg_ts[i], reg_prev_seg[i], 0, 0, 0});
  }
  // The rows that close the session: one per word any segment touched or the image holds, and per register touched, in increasing
  // address order.  `own` (sorted) are the boundary rows of the segment that is being closed (empty for a segment without cycles): an
  // address among them keeps its history; any other is merely named -- found and left as it is, never accessed.  Each row carries the
  // address's initial value: the image's word, or zero.
  void closing_rows(const std::vector<r0h_preflight_bound>& own, std::vector<r0h_preflight_bound>& out) {
    std::vector<uint32_t> pages(vm.page_list);
    std::sort(pages.begin(), pages.end());
    size_t io = 0, im = 0;
    const std::vector<std::pair<uint32_t, uint32_t>>& image = vm.image;
    auto emit = [&](uint32_t addr, uint32_t value, uint32_t seg) {
      while (im < image.size() && image[im].first < addr) im++;
      const bool in_image = im < image.size() && image[im].first == addr;
      r0h_preflight_bound b;
      if (io < own.size() && own[io].addr == addr) b = own[io++];
      else b = r0h_preflight_bound{addr, value, value, 0, seg, 0, 0, 0};
      b.init_value = in_image ? image[im].second : 0u;
      b.flags = in_image ? R0H_BOUND_IMAGE : 0u;
      out.push_back(b);
    };
    for (uint32_t idx : pages) {
      const Page* p = vm.table[idx];
      // the image's words of this page that nothing has touched are named too (the verifier multiplies out the whole image)
      size_t lo = std::lower_bound(image.begin(), image.end(), std::make_pair(idx << 8, 0u)) - image.begin();
      for (uint32_t k = 0; k < PAGE_WORDS; k++) {
        const uint32_t addr = idx << 8 | k;
        while (lo < image.size() && image[lo].first < addr) lo++;
        const bool in_image = lo < image.size() && image[lo].first == addr;
        if (p->seg[k] || in_image) emit(addr, p->w[k], p->seg[k]);
      }
    }
    for (uint32_t i = 0; i < 32; i++)
      if (vm.reg_seg[i]) emit(R0H_REG_BASE + i, vm.x[i], vm.reg_seg[i]);
  }
  void push_segment(uint32_t exit_system, uint32_t exit_user) {
    cur.info.post.pc = vm.pc;
    memory_root(vm, root_now);
    memcpy(cur.info.post.merkle_root, root_now, 32);
    cur.info.pages_in = n_in;
    cur.info.pages_out = n_out;
    cur.info.paging_cycles = paging_cycles();
    cur.info.exit_system = exit_system;
    cur.info.exit_user = exit_user;
  }
  // final: the run ends with this segment -- the session is closed in it if the closing rows fit beside its cycles, in segments
  // of closing rows only (no cycles) after it otherwise
  void end_segment(uint32_t exit_system, uint32_t exit_user, bool final = false) {
    push_segment(exit_system, exit_user);
    cur.info.boundary_rows = (uint32_t)boundary_count();
    if (!(lim.keep_trace && lim.boundary_rows)) {
      vm.segments.push_back(std::move(cur));
      return;
    }
    own_bounds(cur.bounds);
    if (!final) {
      vm.segments.push_back(std::move(cur));
      return;
    }
    std::vector<r0h_preflight_bound> all;
    closing_rows(cur.bounds, all);
    if (cur.info.user_cycles + all.size() <= seg_budget) {
      cur.bounds.swap(all);
      cur.info.boundary_rows = (uint32_t)cur.bounds.size();
      cur.info.closing = 1;
      vm.segments.push_back(std::move(cur));
      return;
    }
    vm.segments.push_back(std::move(cur));
    all.clear();
    closing_rows(std::vector<r0h_preflight_bound>(), all);  // every address now names the segment just pushed, or an earlier one
    for (size_t at = 0; at < all.size(); at += (size_t)seg_budget) {
      begin_segment();
      cur.bounds.assign(all.begin() + at, all.begin() + std::min(all.size(), at + (size_t)seg_budget));
      push_segment(2, 0);
      cur.info.boundary_rows = (uint32_t)cur.bounds.size();
      cur.info.closing = 1;
      vm.segments.push_back(std::move(cur));
    }
  }

  // ---- the accesses of a cycle, each with its timestamp: returns when the same register / word was last touched in this segment
  uint32_t touch_reg(uint32_t i, uint32_t ts) {
    const uint32_t prev = reg_ts[i];
    if (!(reg_mask >> i & 1)) { reg_mask |= 1u << i; reg_first[i] = vm.x[i]; reg_prev_seg[i] = vm.reg_seg[i]; vm.reg_seg[i] = epoch; }
    reg_ts[i] = ts;
    return prev;
  }
  Page* page(uint32_t addr, bool write) {
    Page* p = vm.get(addr >> PAGE_SHIFT);
    if (p->in_epoch != epoch) { p->in_epoch = epoch; n_in++; }
    if (write) {
      if (p->out_epoch != epoch) { p->out_epoch = epoch; n_out++; }
      vm.mark_stale(p, addr >> PAGE_SHIFT);
    }
    return p;
  }
  uint32_t touch_word(Page* p, uint32_t addr, uint32_t ts) {
    if (p->epoch != epoch) { p->epoch = epoch; memset(p->ts, 0, sizeof p->ts); }
    const uint32_t k = (addr & (PAGE_BYTES - 1)) >> 2, prev = p->ts[k];
    if (!prev) { touched.push_back(Touched{addr >> 2, p->w[k], p->seg[k]}); p->seg[k] = epoch; }
    p->ts[k] = ts;
    return prev;
  }
  // the cycle's one memory access (word-aligned address): value before, value after
  void mem_access(uint32_t addr, bool write, uint32_t value, uint32_t* before) {
    Page* p = page(addr, write);
    const uint32_t k = (addr & (PAGE_BYTES - 1)) >> 2;
    const uint32_t prev = touch_word(p, addr, r0h::trace::stamp((uint32_t)cur.info.user_cycles, r0h::trace::ACC_MEM));
    *before = p->w[k];
    if (write) p->w[k] = value;
    if (row) {
      row->mem_addr = addr; row->mem_before = *before; row->mem_after = write ? value : *before;
      row->mem_kind = write ? R0H_MEM_WRITE : R0H_MEM_READ;
      row->prev[3] = prev;
    }
  }
  bool load(uint32_t addr, uint32_t width, bool sign, uint32_t* out) {
    if (addr & (width - 1)) { err = "misaligned load"; return false; }
    if (addr >> ADDRESS_BITS) { err = "load outside the 1 GiB address space"; return false; }
    uint32_t w;
    mem_access(addr & ~3u, false, 0, &w);
    const uint32_t sh = 8 * (addr & 3);
    uint32_t v = width == 4 ? w : (w >> sh) & (width == 1 ? 0xffu : 0xffffu);
    if (sign && width == 1) v = (uint32_t)(int32_t)(int8_t)v;
    if (sign && width == 2) v = (uint32_t)(int32_t)(int16_t)v;
    *out = v;
    return true;
  }
  bool store(uint32_t addr, uint32_t width, uint32_t v) {
    if (addr & (width - 1)) { err = "misaligned store"; return false; }
    if (addr >> ADDRESS_BITS) { err = "store outside the 1 GiB address space"; return false; }
    if (width == 4) { uint32_t old; mem_access(addr, true, v, &old); return true; }
    // a narrow store rewrites part of the word: the new word depends on the old one, read in the same access
    Page* p = page(addr & ~3u, true);
    const uint32_t k = (addr & (PAGE_BYTES - 1)) >> 2, sh = 8 * (addr & 3);
    const uint32_t mask = (width == 1 ? 0xffu : 0xffffu) << sh;
    const uint32_t nw = (p->w[k] & ~mask) | ((v << sh) & mask);
    uint32_t old;
    mem_access(addr & ~3u, true, nw, &old);
    return true;
  }

  // one cycle; returns false when the run ends (halt / pause / error)
  bool step() {
    r0h_vm& m = vm;
    if (m.pc & 3) { err = "misaligned pc"; return false; }
    if (m.pc >> ADDRESS_BITS) { err = "pc outside the 1 GiB address space"; return false; }
    // segment boundary: the next cycle, the pages it may bring in and write back (its own and one more) and the boundary rows of
    // what it may touch for the first time (three registers, one word, the fetched word) must still fit
    const uint64_t worst = (uint64_t)lim.page_in_cycles * 2 + lim.page_out_cycles + (lim.boundary_rows ? 5 : 0);
    if (cur.info.user_cycles + 1 + paging_cycles() + boundary_count() + worst > seg_budget) {
      if (cur.info.user_cycles != 0) {
        end_segment(2, 0);  // SystemSplit
        begin_segment();
      }
      if (1 + worst > seg_budget) { err = "segment limit too small for a single instruction and its pages"; return false; }
    }
    const uint32_t cyc = (uint32_t)cur.info.user_cycles;
    Page* fp = page(m.pc, false);
    const uint32_t insn = fp->w[(m.pc & (PAGE_BYTES - 1)) >> 2];
    const uint32_t fetch_prev = touch_word(fp, m.pc, r0h::trace::stamp(cyc, r0h::trace::ACC_FETCH));
    const uint32_t op = insn & 0x7f, rd = (insn >> 7) & 31, f3 = (insn >> 12) & 7, rs1 = (insn >> 15) & 31, rs2 = (insn >> 20) & 31, f7 = insn >> 25;
    const bool sys = op == 0x73;  // an ecall reads a7 and a0 where another instruction reads x[rs1] and x[rs2]
    const uint32_t r1 = sys ? 17u : rs1, r2 = sys ? 10u : rs2;
    const uint32_t a = m.x[r1], b = m.x[r2];
    const int32_t imm_i = (int32_t)insn >> 20;
    const int32_t imm_s = ((int32_t)(insn & 0xfe000000) >> 20) | (int32_t)((insn >> 7) & 31);
    const int32_t imm_b = ((int32_t)(insn & 0x80000000) >> 19) | (int32_t)((insn & 0x80) << 4) | (int32_t)((insn >> 20) & 0x7e0) | (int32_t)((insn >> 7) & 0x1e);
    const int32_t imm_j = ((int32_t)(insn & 0x80000000) >> 11) | (int32_t)(insn & 0xff000) | (int32_t)((insn >> 9) & 0x800) | (int32_t)((insn >> 20) & 0x7fe);
    uint32_t next = m.pc + 4, wr = 0, wr_reg = rd;
    bool has_wr = false;
    // x0 is a register like any other to the memory argument (nothing ever writes it): every cycle reads two registers
    const uint32_t p0 = touch_reg(r1, r0h::trace::stamp(cyc, r0h::trace::ACC_RS1)), p1 = touch_reg(r2, r0h::trace::stamp(cyc, r0h::trace::ACC_RS2));
    if (lim.keep_trace) {
      cur.rows.emplace_back();  // value-initialised: all zero
      row = &cur.rows.back();
      row->cycle = cyc; row->pc = m.pc; row->insn = insn; row->rs1_value = a; row->rs2_value = b;
      row->prev[0] = p0; row->prev[1] = p1; row->prev[4] = fetch_prev;
    } else {
      row = nullptr;
    }
    auto set = [&](uint32_t v) { wr = v; has_wr = true; };
    bool running = true;
    switch (op) {
      case 0x37: set(insn & 0xfffff000u); break;                       // LUI
      case 0x17: set(m.pc + (insn & 0xfffff000u)); break;              // AUIPC
      case 0x6f: set(m.pc + 4); next = m.pc + (uint32_t)imm_j; break;  // JAL
      case 0x67:                                                       // JALR
        if (f3 != 0) { err = "illegal instruction"; return false; }
        set(m.pc + 4);
        next = (a + (uint32_t)imm_i) & ~1u;
        break;
      case 0x63: {  // branches
        bool take;
        switch (f3) {
          case 0: take = a == b; break;
          case 1: take = a != b; break;
          case 4: take = (int32_t)a < (int32_t)b; break;
          case 5: take = (int32_t)a >= (int32_t)b; break;
          case 6: take = a < b; break;
          case 7: take = a >= b; break;
          default: err = "illegal instruction"; return false;
        }
        if (take) next = m.pc + (uint32_t)imm_b;
        break;
      }
      case 0x03: {  // loads
        uint32_t v;
        const uint32_t addr = a + (uint32_t)imm_i;
        bool ok;
        switch (f3) {
          case 0: ok = load(addr, 1, true, &v); break;
          case 1: ok = load(addr, 2, true, &v); break;
          case 2: ok = load(addr, 4, false, &v); break;
          case 4: ok = load(addr, 1, false, &v); break;
          case 5: ok = load(addr, 2, false, &v); break;
          default: err = "illegal instruction"; return false;
        }
        if (!ok) return false;
        set(v);
        break;
      }
      case 0x23: {  // stores
        const uint32_t addr = a + (uint32_t)imm_s;
        if (f3 > 2) { err = "illegal instruction"; return false; }
        if (!store(addr, 1u << f3, b)) return false;
        break;
      }
      case 0x13: {  // register-immediate
        const uint32_t sh = rs2;
        switch (f3) {
          case 0: set(a + (uint32_t)imm_i); break;
          case 2: set((int32_t)a < imm_i); break;
          case 3: set(a < (uint32_t)imm_i); break;
          case 4: set(a ^ (uint32_t)imm_i); break;
          case 6: set(a | (uint32_t)imm_i); break;
          case 7: set(a & (uint32_t)imm_i); break;
          case 1: if (f7 != 0) { err = "illegal instruction"; return false; } set(a << sh); break;
          case 5:
            if (f7 == 0) set(a >> sh);
            else if (f7 == 0x20) set((uint32_t)((int32_t)a >> sh));
            else { err = "illegal instruction"; return false; }
            break;
        }
        break;
      }
      case 0x33: {  // register-register, incl. the M extension
        if (f7 == 0x01) {
          const int64_t sa = (int32_t)a, sb = (int32_t)b;
          switch (f3) {
            case 0: set(a * b); break;                                                   // MUL
            case 1: set((uint32_t)((uint64_t)(sa * sb) >> 32)); break;                   // MULH
            case 2: set((uint32_t)((uint64_t)(sa * (int64_t)(uint64_t)b) >> 32)); break; // MULHSU
            case 3: set((uint32_t)(((uint64_t)a * b) >> 32)); break;                     // MULHU
            case 4: set(b == 0 ? 0xffffffffu : (a == 0x80000000u && b == 0xffffffffu) ? a : (uint32_t)((int32_t)a / (int32_t)b)); break;  // DIV
            case 5: set(b == 0 ? 0xffffffffu : a / b); break;                            // DIVU
            case 6: set(b == 0 ? a : (a == 0x80000000u && b == 0xffffffffu) ? 0u : (uint32_t)((int32_t)a % (int32_t)b)); break;          // REM
            case 7: set(b == 0 ? a : a % b); break;                                      // REMU
          }
        } else if (f7 == 0x00 || f7 == 0x20) {
          const bool alt = f7 == 0x20;
          if (alt && f3 != 0 && f3 != 5) { err = "illegal instruction"; return false; }
          switch (f3) {
            case 0: set(alt ? a - b : a + b); break;
            case 1: set(a << (b & 31)); break;
            case 2: set((int32_t)a < (int32_t)b); break;
            case 3: set(a < b); break;
            case 4: set(a ^ b); break;
            case 5: set(alt ? (uint32_t)((int32_t)a >> (b & 31)) : a >> (b & 31)); break;
            case 6: set(a | b); break;
            case 7: set(a & b); break;
          }
        } else { err = "illegal instruction"; return false; }
        break;
      }
      case 0x0f: break;  // FENCE: a no-op for a single hart
      case 0x73: {
        if (insn != 0x00000073u) { err = insn == 0x00100073u ? "ebreak" : "illegal instruction"; return false; }
        const uint32_t fn = a, a0 = b, a1 = m.x[11];
        switch (fn) {
          case 0: exit_kind = R0H_VM_HALTED; exit_code = a0; running = false; break;
          case 4: exit_kind = R0H_VM_PAUSED; exit_code = a0; running = false; break;
          case 1:
          case 2: {
            if (a0 & 3) { err = fn == 1 ? "READ_WORDS: misaligned destination" : "COMMIT: misaligned source"; return false; }
            if (a0 >> ADDRESS_BITS) { err = "ecall buffer outside the 1 GiB address space"; return false; }
            wr_reg = 11;
            if (a1 == 0) {  // nothing (left) to move: fall through; a1 is written all the same (the cycle's register write)
              if (m.io_active && fn == 1) m.input_pos = std::min(m.input.size(), m.io_base + m.io_total);
              m.io_active = false;
              set(0);
              break;
            }
            if (!m.io_active) {
              if (a1 > MAX_IO_WORDS) { err = fn == 1 ? "READ_WORDS: count too large" : "COMMIT: count too large"; return false; }
              if (((uint64_t)a0 + 4ull * a1) >> ADDRESS_BITS) { err = "ecall buffer outside the 1 GiB address space"; return false; }
              m.io_active = true;
              m.io_total = a1;
              m.io_base = fn == 1 ? m.input_pos : 0;
              if (fn == 2) {  // the journal is a window of memory: word i of it lives at R0H_JOURNAL_BASE + 4 i
                if (a0 < R0H_JOURNAL_BASE || (uint64_t)a0 + 4ull * a1 > (uint64_t)R0H_JOURNAL_BASE + MAX_JOURNAL_BYTES) { err = "COMMIT: the source lies outside the journal window"; return false; }
              }
            }
            if (a1 > m.io_total) { err = "ecall: a1 grew during a transfer"; return false; }
            const uint32_t idx = a1 - 1;  // the words go from the back of the buffer to its front
            if (fn == 1) {
              uint32_t old;
              mem_access(a0 + 4 * idx, true, m.io_base + idx < m.input.size() ? m.input[m.io_base + idx] : 0u, &old);
            } else {
              uint32_t w;
              if (a0 < R0H_JOURNAL_BASE) { err = "COMMIT: the source lies outside the journal window"; return false; }
              const size_t j = (size_t)(a0 - R0H_JOURNAL_BASE) / 4 + idx;
              if (4 * (j + 1) > MAX_JOURNAL_BYTES) { err = "COMMIT: the journal exceeds 2^28 bytes"; return false; }
              if (m.journal_seen.size() <= j) { m.journal_seen.resize(j + 1, 0); m.journal.resize(4 * (j + 1), 0); }
              if (m.journal_seen[j]) { err = "COMMIT: a journal word is committed twice"; return false; }
              m.journal_seen[j] = 1;
              mem_access(a0 + 4 * idx, false, 0, &w);
              for (uint32_t i = 0; i < 4; i++) m.journal[4 * j + i] = (uint8_t)(w >> (8 * i));
            }
            set(idx);
            next = m.pc;
            break;
          }
          case 3: wr_reg = 10; set((uint32_t)m.cycles); break;
          default: err = "unknown ecall function"; return false;
        }
        break;
      }
      default: err = "illegal instruction"; return false;
    }
    if (has_wr && wr_reg != 0) {
      const uint32_t p2 = touch_reg(wr_reg, r0h::trace::stamp(cyc, r0h::trace::ACC_RD));
      if (row) { row->rd = wr_reg; row->rd_before = m.x[wr_reg]; row->rd_after = wr; row->prev[2] = p2; }
      m.x[wr_reg] = wr;
    }
    if (row) row->next_pc = next;
    m.pc = next;
    m.cycles++;
    cur.info.user_cycles++;
    return running;
  }

  // runs until one more segment is complete (or the run ends); returns an error string or nullptr
  const char* run_segment(bool* finished) {
    const size_t before = vm.segments.size();
    *finished = false;
    while (vm.segments.size() == before) {
      if (lim.max_cycles && executed >= lim.max_cycles) { end_segment(2, 2, true); *finished = true; return nullptr; }  // SessionLimit
      const bool running = step();
      if (err) return r0h::make_error("guest trap at pc %#x after %llu cycles: %s", vm.pc, (unsigned long long)vm.cycles, err);
      executed++;
      if (!running) {
        for (uint8_t seen : vm.journal_seen)
          if (!seen) return r0h::make_error("the guest's commits leave a hole in the journal window");
        end_segment(exit_kind == R0H_VM_HALTED ? 0 : 1, exit_code, true);  // Halted(code) / Paused(code)
        *finished = true;
        return nullptr;
      }
    }
    return nullptr;
  }
};
}  // namespace

r0h_vm::~r0h_vm() {
  if (!table) return;
  for (uint32_t idx : page_list) free(table[idx]);
  free(table);
}

namespace r0h {
namespace trace {
const Tables& trace_tables() {
  static const Tables T = [] {
    Tables t;
    memset(&t, 0, sizeof t);
    for (uint32_t i = 1; i < 32; i++) t.inv_small[i] = inv(enc(i));
    return t;
  }();
  return T;
}
// the names tools/gen_circuit.py gives the columns (TRACE_COLUMNS), built the same way
const char* column_name(uint32_t column) {
  static const std::vector<std::string> names = [] {
    std::vector<std::string> n;
    auto run = [&](const char* stem, int count) { for (int k = 0; k < count; k++) n.push_back(stem + std::to_string(k)); };
    for (const char* s : {"live", "bnd", "cycle", "pc", "next_pc"}) n.push_back(s);
    for (const char* s : {"lui", "auipc", "jal", "jalr", "branch", "load", "store", "imm", "op", "system"}) n.push_back(std::string("opc_") + s);
    for (int k = 1; k < 8; k++) n.push_back("f3_" + std::to_string(k));
    for (const char* s : {"alu", "rd0", "rdA", "rdB", "r10", "r1A", "r1B", "r20", "r2A", "r2B", "b25", "f7A", "f7B", "b30", "b31",
                          "rs1_lo", "rs1_hi", "dl0", "dh0", "rs2_lo", "rs2_hi", "dl1", "dh1",
                          "zrd", "inv_rd", "act2", "old_lo", "old_hi", "dl2", "dh2",
                          "mem_wr", "top", "addr3", "before_lo", "before_hi", "after_lo", "after_hi", "p3", "dl3", "dh3", "dl4", "dh4"})
      n.push_back(s);
    run("u", 4);
    for (int k = 1; k < 4; k++) n.push_back("v" + std::to_string(k));
    run("a", 4);
    n.push_back("su");
    n.push_back("sv");
    run("sh", 5);
    for (const char* s : {"vrd", "vrb", "z_hi", "ob0", "ob1", "zq", "w_lo", "w_hi", "aux0", "aux1",
                          "res_lo", "res_hi", "c0", "c1", "lt", "eq", "zinv", "sb", "sgn", "p8", "sx", "sm"})
      n.push_back(s);
    run("mb", 4);
    for (const char* s : {"ce0", "ce1a", "ce1b", "ce2", "cb1", "cb2", "cband", "c3", "dv", "ovf", "k0", "a31", "f0", "f1", "f2", "fn_cyc", "cact", "fimg", "m16", "mand"})
      n.push_back(s);
    return n;
  }();
  return column < names.size() ? names[column].c_str() : nullptr;
}
}  // namespace trace
}  // namespace r0h

using namespace r0h;

extern "C" {

const char* r0h_vm_new(r0h_vm** out) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(out, "r0h_vm_new: NULL argument");
  std::unique_ptr<r0h_vm> vm(new r0h_vm);
  R0H_REQUIRE(vm->table, "r0h_vm_new: out of memory");
  *out = vm.release();
  return nullptr;
  R0H_GUARD_END
}
const char* r0h_vm_free(r0h_vm* vm) {
  delete vm;
  return nullptr;
}

const char* r0h_vm_load(r0h_vm* vm, uint32_t addr, const uint32_t* words, size_t n) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(vm && (words || !n), "r0h_vm_load: NULL argument");
  R0H_REQUIRE((addr & 3) == 0 && (uint64_t)addr + 4 * (uint64_t)n <= ((uint64_t)1 << 32), "r0h_vm_load: [%#x, +%zu words) is misaligned or leaves the address space", addr, n);
  R0H_REQUIRE(!vm->run && vm->segments.empty(), "r0h_vm_load: the run has begun (what is loaded before it is the program image)");
  for (size_t i = 0; i < n; i++) {
    const uint32_t a = addr + 4 * (uint32_t)i;
    Page* p = vm->get(a >> PAGE_SHIFT);
    p->w[(a & (PAGE_BYTES - 1)) >> 2] = words[i];
    vm->mark_stale(p, a >> PAGE_SHIFT);
    vm->image.emplace_back(a >> 2, words[i]);
  }
  vm->image_sorted = false;
  return nullptr;
  R0H_GUARD_END
}

// ELF32 little-endian RISC-V executable: PT_LOAD segments into memory, entry point into pc (risc0-binfmt `Program::load_elf`)
const char* r0h_vm_load_elf(r0h_vm* vm, const uint8_t* elf, size_t n) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(vm && elf, "r0h_vm_load_elf: NULL argument");
  auto u16 = [&](size_t o) { return (uint32_t)elf[o] | (uint32_t)elf[o + 1] << 8; };
  auto u32 = [&](size_t o) { return u16(o) | u16(o + 2) << 16; };
  R0H_REQUIRE(n >= 52 && !memcmp(elf, "\x7f" "ELF", 4), "ELF: bad magic");
  R0H_REQUIRE(elf[4] == 1 && elf[5] == 1, "ELF: not 32-bit little-endian");
  R0H_REQUIRE(u16(16) == 2 && u16(18) == 243, "ELF: not a RISC-V executable (e_type %u, e_machine %u)", u16(16), u16(18));
  const uint32_t entry = u32(24), phoff = u32(28), phentsize = u16(42), phnum = u16(44);
  R0H_REQUIRE((entry & 3) == 0, "ELF: misaligned entry point");
  R0H_REQUIRE(phentsize >= 32 && (uint64_t)phoff + (uint64_t)phentsize * phnum <= n, "ELF: program headers outside the file");
  for (uint32_t i = 0; i < phnum; i++) {
    const size_t ph = phoff + (size_t)i * phentsize;
    if (u32(ph) != 1) continue;  // PT_LOAD
    const uint32_t off = u32(ph + 4), vaddr = u32(ph + 8), filesz = u32(ph + 16), memsz = u32(ph + 20);
    R0H_REQUIRE((vaddr & 3) == 0 && filesz <= memsz && (uint64_t)off + filesz <= n && (uint64_t)vaddr + memsz <= ((uint64_t)1 << 32), "ELF: segment %u is malformed", i);
    std::vector<uint32_t> words((filesz + 3) / 4, 0);
    for (uint32_t b = 0; b < filesz; b++) words[b / 4] |= (uint32_t)elf[off + b] << (8 * (b % 4));
    R0H_TRY(r0h_vm_load(vm, vaddr, words.data(), words.size()));
    // [filesz, memsz) is .bss: memory nothing has touched reads as zero already, so only what an earlier segment put there is
    // cleared -- a header may claim hundreds of megabytes, none of which need exist
    const uint64_t z0 = (uint64_t)vaddr + 4 * (uint64_t)words.size(), z1 = (uint64_t)vaddr + memsz;
    for (uint32_t idx : vm->page_list) {
      const uint64_t base = (uint64_t)idx << PAGE_SHIFT;
      if (base + PAGE_BYTES <= z0 || base >= z1) continue;
      Page* p = vm->table[idx];
      for (uint32_t w = 0; w < PAGE_WORDS; w++)
        if (base + 4 * w >= z0 && base + 4 * w < z1 && p->w[w]) { p->w[w] = 0; vm->image.emplace_back((uint32_t)((base + 4 * w) >> 2), 0u); }
      vm->mark_stale(p, idx);
    }
  }
  vm->pc = entry;
  return nullptr;
  R0H_GUARD_END
}

// risc0-binfmt `compute_image_id(elf)` in this library's terms: the digest of the SystemState a run of this ELF starts from (entry pc,
// Merkle root of the loaded pages) -- what r0h_prove_elf returns as image id and r0h_receipt_verify holds a receipt against
const char* r0h_compute_image_id(const uint8_t* elf, size_t n, uint8_t image_id_out[32]) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(elf && image_id_out, "r0h_compute_image_id: NULL argument");
  r0h_vm* vm = nullptr;
  R0H_TRY(r0h_vm_new(&vm));
  struct Guard { r0h_vm* v; ~Guard() { r0h_vm_free(v); } } guard{vm};
  R0H_TRY(r0h_vm_load_elf(vm, elf, n));
  r0h_system_state st;
  memset(&st, 0, sizeof st);
  st.pc = vm->pc;
  initial_root(*vm, st.merkle_root);
  system_state_digest(st, image_id_out);
  return nullptr;
  R0H_GUARD_END
}

const char* r0h_vm_set_input(r0h_vm* vm, const uint32_t* words, size_t n) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(vm && (words || !n), "r0h_vm_set_input: NULL argument");
  vm->input.assign(words, words + n);
  vm->input_pos = 0;
  return nullptr;
  R0H_GUARD_END
}
const char* r0h_vm_set_pc(r0h_vm* vm, uint32_t pc) {
  R0H_REQUIRE(vm && (pc & 3) == 0, "r0h_vm_set_pc: NULL vm or misaligned pc");
  vm->pc = pc;
  return nullptr;
}
uint32_t r0h_vm_reg(const r0h_vm* vm, uint32_t i) { return vm && i < 32 ? vm->x[i] : 0; }
uint32_t r0h_vm_pc(const r0h_vm* vm) { return vm ? vm->pc : 0; }
const char* r0h_vm_set_reg(r0h_vm* vm, uint32_t i, uint32_t v) {
  R0H_REQUIRE(vm && i < 32, "r0h_vm_set_reg: bad register");
  if (i) vm->x[i] = v;
  return nullptr;
}
const char* r0h_vm_read(const r0h_vm* vm, uint32_t addr, uint32_t* words, size_t n) {
  R0H_REQUIRE(vm && words && (addr & 3) == 0, "r0h_vm_read: NULL argument or misaligned address");
  for (size_t i = 0; i < n; i++) {
    const uint32_t a = addr + 4 * (uint32_t)i;
    const Page* p = vm->find(a >> PAGE_SHIFT);
    words[i] = p ? p->w[(a & (PAGE_BYTES - 1)) >> 2] : 0u;
  }
  return nullptr;
}

const char* r0h_vm_run_segment(r0h_vm* vm, const r0h_vm_limits* limits, int* finished_out, int* exit_kind, uint32_t* exit_code) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(vm && limits && finished_out, "r0h_vm_run_segment: NULL argument");
  R0H_REQUIRE(limits->segment_po2 >= 6 && limits->segment_po2 <= 24, "r0h_vm_run: segment_po2 %u outside [6, 24]", limits->segment_po2);
  R0H_REQUIRE(!vm->finished, "r0h_vm_run: this machine has already run to its end (one run per r0h_vm: load a new one)");
  if (!vm->run) {
    vm->run.reset(new Run(*vm, *limits));
    vm->run->begin_segment();
  }
  bool finished = false;
  const char* err = vm->run->run_segment(&finished);
  if (err || finished) vm->finished = true;
  if (err) { vm->run.reset(); return err; }
  *finished_out = finished ? 1 : 0;
  if (finished) {
    if (exit_kind) *exit_kind = vm->run->exit_kind;
    if (exit_code) *exit_code = vm->run->exit_code;
    vm->run.reset();
  }
  return nullptr;
  R0H_GUARD_END
}

const char* r0h_vm_run(r0h_vm* vm, const r0h_vm_limits* limits, int* exit_kind, uint32_t* exit_code) {
  R0H_REQUIRE(vm && limits && exit_kind && exit_code, "r0h_vm_run: NULL argument");
  *exit_kind = R0H_VM_LIMIT;
  *exit_code = 0;
  for (int finished = 0; !finished;) R0H_TRY(r0h_vm_run_segment(vm, limits, &finished, exit_kind, exit_code));
  return nullptr;
}

}  // extern "C"
namespace r0h {
// the sponge's blocks for an image (tools/image_circuit.py `blocks`), Montgomery words: 16 per block
void image_stream(const std::vector<std::pair<uint32_t, uint32_t>>& image, std::vector<uint32_t>& out) {
  const size_t n_blocks = image.empty() ? 1 : (image.size() + 3) / 4;
  out.assign(16 * n_blocks, 0u);
  for (size_t k = 0; k < image.size(); k++) {
    uint32_t* blk = &out[16 * (k / 4)];
    const size_t j = k % 4;
    blk[3 * j] = enc(image[k].first);
    blk[3 * j + 1] = enc(image[k].second & 0xffffu);
    blk[3 * j + 2] = enc(image[k].second >> 16);
    blk[12] = enc((2u << j) - 1u);  // the tuples there are a prefix of the block: the mask of the last one written stands
  }
}
void image_digest(const std::vector<std::pair<uint32_t, uint32_t>>& image, uint32_t digest[8]) {
  std::vector<uint32_t> stream;
  image_stream(image, stream);
  std::unique_ptr<P2Consts> k(new P2Consts);
  p2_default_host(*k);
  p2_hash_elems_host(*k, stream.data(), stream.size(), digest);
}
// what a verifier needs of an ELF: the image as (word index, word) in address order, the entry point, the image id
const char* elf_image(const uint8_t* elf, size_t n, std::vector<std::pair<uint32_t, uint32_t>>& image, uint32_t* entry, uint8_t image_id[32]) {
  r0h_vm* vm = nullptr;
  R0H_TRY(r0h_vm_new(&vm));
  struct Guard { r0h_vm* v; ~Guard() { r0h_vm_free(v); } } guard{vm};
  R0H_TRY(r0h_vm_load_elf(vm, elf, n));
  sort_image(*vm);
  image = vm->image;
  *entry = vm->pc;
  r0h_system_state st;
  memset(&st, 0, sizeof st);
  st.pc = vm->pc;
  initial_root(*vm, st.merkle_root);
  system_state_digest(st, image_id);
  return nullptr;
}
// the rows of segment i change hands (session.cpp hands them to the prover while the guest runs on)
void vm_take_trace(r0h_vm* vm, size_t i, std::vector<r0h_preflight_row>& rows, std::vector<r0h_preflight_bound>& bounds) {
  rows.swap(vm->segments[i].rows);
  bounds.swap(vm->segments[i].bounds);
}
// ... and come back empty-handed but with their memory, for a later segment to fill (call from the thread that runs the machine)
void vm_recycle_trace(r0h_vm* vm, std::vector<r0h_preflight_row>& rows, std::vector<r0h_preflight_bound>& bounds) {
  if (rows.capacity()) { vm->spare_rows.emplace_back(); vm->spare_rows.back().swap(rows); }
  if (bounds.capacity()) { vm->spare_bounds.emplace_back(); vm->spare_bounds.back().swap(bounds); }
}
// the unused row buffers leave the machine (a session keeps them, page-locked, for its context's next run)
void vm_take_spares(r0h_vm* vm, std::vector<std::vector<r0h_preflight_row>>& rows, std::vector<std::vector<r0h_preflight_bound>>& bounds) {
  for (auto& r : vm->spare_rows) { r.clear(); rows.emplace_back(); rows.back().swap(r); }
  for (auto& b : vm->spare_bounds) { b.clear(); bounds.emplace_back(); bounds.back().swap(b); }
  vm->spare_rows.clear();
  vm->spare_bounds.clear();
  for (auto& s : vm->segments) {  // ... and those of segments nobody took
    if (s.rows.capacity()) { s.rows.clear(); rows.emplace_back(); rows.back().swap(s.rows); }
    if (s.bounds.capacity()) { s.bounds.clear(); bounds.emplace_back(); bounds.back().swap(s.bounds); }
  }
}
}  // namespace r0h
extern "C" {

const char* r0h_vm_release_trace(r0h_vm* vm, size_t i) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(vm && i < vm->segments.size(), "r0h_vm_release_trace: no such segment");
  std::vector<r0h_preflight_row>().swap(vm->segments[i].rows);
  std::vector<r0h_preflight_bound>().swap(vm->segments[i].bounds);
  return nullptr;
  R0H_GUARD_END
}

size_t r0h_vm_n_segments(const r0h_vm* vm) { return vm ? vm->segments.size() : 0; }
uint64_t r0h_vm_cycles(const r0h_vm* vm) { return vm ? vm->cycles : 0; }

const char* r0h_vm_segment_info(const r0h_vm* vm, size_t i, r0h_vm_segment* out) {
  R0H_REQUIRE(vm && out, "r0h_vm_segment_info: NULL argument");
  R0H_REQUIRE(i < vm->segments.size(), "r0h_vm_segment_info: segment %zu of %zu", i, vm->segments.size());
  *out = vm->segments[i].info;
  return nullptr;
}
const char* r0h_vm_preflight(const r0h_vm* vm, size_t i, const r0h_preflight_row** rows, size_t* n) {
  R0H_REQUIRE(vm && rows && n, "r0h_vm_preflight: NULL argument");
  R0H_REQUIRE(i < vm->segments.size(), "r0h_vm_preflight: segment %zu of %zu", i, vm->segments.size());
  *rows = vm->segments[i].rows.data();
  *n = vm->segments[i].rows.size();
  return nullptr;
}
const char* r0h_vm_boundary(const r0h_vm* vm, size_t i, const r0h_preflight_bound** rows, size_t* n) {
  R0H_REQUIRE(vm && rows && n, "r0h_vm_boundary: NULL argument");
  R0H_REQUIRE(i < vm->segments.size(), "r0h_vm_boundary: segment %zu of %zu", i, vm->segments.size());
  *rows = vm->segments[i].bounds.data();
  *n = vm->segments[i].bounds.size();
  return nullptr;
}
const char* r0h_trace_column_name(uint32_t column) { return column < trace::N_COLS ? trace::column_name(column) : nullptr; }

// The DATA group of the trace circuit on the host (the reference tests compare r0h_trace_witgen's device kernel with): cycles first,
// then the boundary rows, blank rows to the end -- csrc/trace.hpp holds the expansion both sides compile.  The two multiplicity
// columns are left zero: r0h_logup_multiplicities_host counts the lookups (from the circuit's own description of them).
const char* r0h_vm_trace_witness(const r0h_vm* vm, size_t i, uint32_t po2, uint32_t* data_out, uint32_t globals_out[R0H_TRACE_GLOBALS]) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(vm && data_out && globals_out, "r0h_vm_trace_witness: NULL argument");
  R0H_REQUIRE(i < vm->segments.size(), "r0h_vm_trace_witness: segment %zu of %zu", i, vm->segments.size());
  const std::vector<r0h_preflight_row>& rows = vm->segments[i].rows;
  const std::vector<r0h_preflight_bound>& bounds = vm->segments[i].bounds;
  const r0h_vm_segment& info = vm->segments[i].info;
  R0H_REQUIRE(!rows.empty() || (info.user_cycles == 0 && !bounds.empty()), "r0h_vm_trace_witness: segment %zu has no preflight rows (run with keep_trace)", i);
  R0H_REQUIRE(po2 >= R0H_TRACE_MIN_PO2 && po2 <= R0H_TRACE_MAX_PO2, "r0h_vm_trace_witness: po2 %u outside [%u, %u] (the lookup tables have 2^16 rows)", po2,
              (unsigned)R0H_TRACE_MIN_PO2, (unsigned)R0H_TRACE_MAX_PO2);
  R0H_REQUIRE(rows.size() + bounds.size() <= ((size_t)1 << po2), "r0h_vm_trace_witness: %zu cycles and %zu boundary rows do not fit 2^%u", rows.size(), bounds.size(), po2);
  const size_t n = (size_t)1 << po2;
  const trace::Tables& T = trace::trace_tables();
  memset(data_out, 0, (size_t)R0H_TRACE_COLUMNS * n * 4);  // the Montgomery form of 0 is 0
  for (size_t r = 0; r < n; r++) {
    auto put = [&](uint32_t col, uint32_t v) { data_out[(size_t)col * n + r] = enc(v); };
    auto raw = [&](uint32_t col, uint32_t w) { data_out[(size_t)col * n + r] = w; };
    if (r < rows.size()) {
      R0H_REQUIRE(rows[r].pc < P && rows[r].next_pc < P, "r0h_vm_trace_witness: pc %#x is not below p: the trace circuit carries a pc as one field element", rows[r].pc);
      trace::live_row(rows[r], T, put, raw);
    } else if (r < rows.size() + bounds.size()) {
      const size_t j = r - rows.size();
      trace::bound_row(bounds[j], j ? bounds[j - 1].addr : 0xffffffffu, info.index + 1, info.closing != 0, T, put, raw);
    } else {
      trace::blank_row(T, put, raw);
    }
  }
  trace::trace_globals(rows.data(), rows.size(), bounds.data(), bounds.size(), info.index + 1, info.closing != 0, info.pre.pc, globals_out);
  return nullptr;
  R0H_GUARD_END
}
const char* r0h_vm_journal(const r0h_vm* vm, const uint8_t** bytes, size_t* n) {
  R0H_REQUIRE(vm && bytes && n, "r0h_vm_journal: NULL argument");
  *bytes = vm->journal.data();
  *n = vm->journal.size();
  return nullptr;
}
// the receipt claim of segment i (csrc/claim.cpp): pre/post system states and exit code from the run, the output digest on the last one
const char* r0h_vm_segment_claim(const r0h_vm* vm, size_t i, r0h_receipt_claim* out) {
  R0H_REQUIRE(vm && out, "r0h_vm_segment_claim: NULL argument");
  R0H_REQUIRE(i < vm->segments.size(), "r0h_vm_segment_claim: segment %zu of %zu", i, vm->segments.size());
  const r0h_vm_segment& s = vm->segments[i].info;
  memset(out, 0, sizeof *out);
  out->pre = s.pre;
  out->post = s.post;
  out->exit_system = s.exit_system;
  out->exit_user = s.exit_user;
  // the segment that ends the run (Halted / Paused) carries the journal's digest -- the last one, unless segments of closing rows follow it
  if (vm->finished && s.exit_system <= 1) R0H_TRY(r0h_output_digest(vm->journal.data(), vm->journal.size(), nullptr, out->output_digest));
  return nullptr;
}

}  // extern "C"
