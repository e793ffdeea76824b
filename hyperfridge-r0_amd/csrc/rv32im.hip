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
//   * the cycle model is one cycle per instruction plus a flat charge per page first touched (paged in) and per page dirtied (paged
//     out) in a segment; risc0's rv32im-v2 charges differ and are not reproducible from the reference;
//   * the state digest is risc0-binfmt's SystemState{pc, merkle_root} (csrc/claim.hip) with merkle_root = a SHA-256 binary Merkle
//     tree over the 1 KiB pages of the 32-bit address space, all-zero subtrees folded (risc0's image id is also a page Merkle root;
//     its exact tree shape and tags are not pinned here).
// No guest ELF exists in the reference (only sources: methods/guest/src/main.rs; the ELF is built by `risc0_build::embed_methods()`,
// methods/build.rs:2, which needs the Rust toolchain), so the camt53 trace itself still cannot be produced.
//
// ecall ABI (a7 = x17 selects; arguments a0.., result in a0):
//   0 HALT        a0 = exit code                        -- ends the run (ExitCode::Halted(a0))
//   1 READ_WORDS  a0 = destination, a1 = word count     -- the next words of the input stream (ExecutorEnv frames), zero past its end
//   2 COMMIT      a0 = source, a1 = byte count          -- appends bytes to the journal (`env::commit`)
//   3 CYCLES                                            -- a0 = cycles executed so far (`env::cycle_count()`)
//   4 PAUSE       a0 = exit code                        -- ends the run resumably (ExitCode::Paused(a0))
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <array>
#include <iterator>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <vector>

#include "../../include/r0hip.h"
#include "internal.hpp"
#include "receipt_types.hpp"

namespace {
constexpr uint32_t PAGE_BYTES = 1024, PAGE_WORDS = PAGE_BYTES / 4, PAGE_SHIFT = 10, N_PAGE_BITS = 32 - PAGE_SHIFT;  // 2^22 pages

struct Segment {
  r0h_vm_segment info;
  std::vector<r0h_preflight_row> rows;
};
}  // namespace

struct r0h_vm {
  uint32_t x[32] = {0};
  uint32_t pc = 0;
  std::map<uint32_t, std::vector<uint32_t>> pages;  // page index -> PAGE_WORDS words; absent = all zero
  std::vector<uint32_t> input;
  size_t input_pos = 0;
  std::vector<uint8_t> journal;
  uint64_t cycles = 0;  // user cycles (instructions) over the whole run
  bool finished = false; // a run ended in HALT / PAUSE / the cycle limit: its segments are final
  std::vector<Segment> segments;
  // hashing caches
  uint8_t zero_level[N_PAGE_BITS + 1][32];
  bool zero_ready = false;
};

namespace {
using r0h::sha256;

void page_hash(const std::vector<uint32_t>& w, uint8_t out[32]) { sha256(w.data(), PAGE_BYTES, out); }  // little-endian words = the bytes of memory
void node_hash(const uint8_t a[32], const uint8_t b[32], uint8_t out[32]) {
  uint8_t cat[64];
  memcpy(cat, a, 32);
  memcpy(cat + 32, b, 32);
  sha256(cat, 64, out);
}
void ensure_zero_levels(r0h_vm& vm) {
  if (vm.zero_ready) return;
  const std::vector<uint32_t> z(PAGE_WORDS, 0);
  page_hash(z, vm.zero_level[0]);
  for (uint32_t l = 1; l <= N_PAGE_BITS; l++) node_hash(vm.zero_level[l - 1], vm.zero_level[l - 1], vm.zero_level[l]);
  vm.zero_ready = true;
}
// Merkle root over all 2^22 pages with all-zero subtrees taken from the table: cost ~ (non-zero pages) x 22 hashes
void memory_root(r0h_vm& vm, uint8_t out[32]) {
  ensure_zero_levels(vm);
  std::map<uint32_t, std::array<uint8_t, 32>> level;
  for (const auto& kv : vm.pages) {
    bool nz = false;
    for (uint32_t w : kv.second) nz |= w != 0;
    if (!nz) continue;
    std::array<uint8_t, 32> h;
    page_hash(kv.second, h.data());
    level[kv.first] = h;
  }
  for (uint32_t l = 0; l < N_PAGE_BITS; l++) {
    std::map<uint32_t, std::array<uint8_t, 32>> up;
    for (auto it = level.begin(); it != level.end();) {
      const uint32_t parent = it->first >> 1;
      const uint8_t *left = vm.zero_level[l], *right = vm.zero_level[l];
      if (it->first & 1) { right = it->second.data(); ++it; }
      else {
        left = it->second.data();
        auto nx = std::next(it);
        if (nx != level.end() && nx->first == (it->first | 1)) { right = nx->second.data(); it = std::next(nx); }
        else ++it;
      }
      std::array<uint8_t, 32> h;
      node_hash(left, right, h.data());
      up[parent] = h;
    }
    level.swap(up);
  }
  if (level.empty()) memcpy(out, vm.zero_level[N_PAGE_BITS], 32);
  else memcpy(out, level.begin()->second.data(), 32);
}

struct Run {
  r0h_vm& vm;
  const r0h_vm_limits& lim;
  Segment cur;
  std::set<uint32_t> touched, dirtied;  // pages of the current segment
  uint64_t seg_budget;
  const char* err = nullptr;
  r0h_preflight_row* row = nullptr;

  Run(r0h_vm& v, const r0h_vm_limits& l) : vm(v), lim(l), seg_budget((uint64_t)1 << l.segment_po2) {}

  uint64_t paging_cycles() const { return (uint64_t)touched.size() * lim.page_in_cycles + (uint64_t)dirtied.size() * lim.page_out_cycles; }

  void begin_segment() {
    cur = Segment();
    memset(&cur.info, 0, sizeof cur.info);
    cur.info.index = (uint32_t)vm.segments.size();
    cur.info.pre.pc = vm.pc;
    memory_root(vm, cur.info.pre.merkle_root);
    touched.clear();
    dirtied.clear();
  }
  void end_segment(uint32_t exit_system, uint32_t exit_user) {
    cur.info.post.pc = vm.pc;
    memory_root(vm, cur.info.post.merkle_root);
    cur.info.pages_in = (uint32_t)touched.size();
    cur.info.pages_out = (uint32_t)dirtied.size();
    cur.info.paging_cycles = paging_cycles();
    cur.info.exit_system = exit_system;
    cur.info.exit_user = exit_user;
    vm.segments.push_back(std::move(cur));
  }

  std::vector<uint32_t>& page(uint32_t addr, bool write) {
    const uint32_t idx = addr >> PAGE_SHIFT;
    touched.insert(idx);
    if (write) dirtied.insert(idx);
    auto it = vm.pages.find(idx);
    if (it == vm.pages.end()) it = vm.pages.emplace(idx, std::vector<uint32_t>(PAGE_WORDS, 0)).first;
    return it->second;
  }
  uint32_t load_word(uint32_t addr) { return page(addr, false)[(addr & (PAGE_BYTES - 1)) >> 2]; }
  void store_word(uint32_t addr, uint32_t v) { page(addr, true)[(addr & (PAGE_BYTES - 1)) >> 2] = v; }
  void note_mem(uint32_t addr, uint32_t before, uint32_t after, uint32_t kind) {
    if (!row) return;
    row->mem_addr = addr & ~3u; row->mem_before = before; row->mem_after = after; row->mem_kind = kind;
  }
  bool load(uint32_t addr, uint32_t width, bool sign, uint32_t* out) {
    if (addr & (width - 1)) { err = "misaligned load"; return false; }
    const uint32_t w = load_word(addr & ~3u), sh = 8 * (addr & 3);
    uint32_t v = width == 4 ? w : (w >> sh) & (width == 1 ? 0xffu : 0xffffu);
    if (sign && width == 1) v = (uint32_t)(int32_t)(int8_t)v;
    if (sign && width == 2) v = (uint32_t)(int32_t)(int16_t)v;
    note_mem(addr, w, w, R0H_MEM_READ);
    *out = v;
    return true;
  }
  bool store(uint32_t addr, uint32_t width, uint32_t v) {
    if (addr & (width - 1)) { err = "misaligned store"; return false; }
    const uint32_t old = load_word(addr & ~3u), sh = 8 * (addr & 3);
    const uint32_t mask = width == 4 ? 0xffffffffu : (width == 1 ? 0xffu : 0xffffu) << sh;
    const uint32_t nw = (old & ~mask) | ((v << sh) & mask);
    store_word(addr & ~3u, nw);
    note_mem(addr, old, nw, R0H_MEM_WRITE);
    return true;
  }

  // one instruction; returns false when the run ends (halt / pause / error)
  bool step(int* exit_kind, uint32_t* exit_code) {
    r0h_vm& m = vm;
    if (m.pc & 3) { err = "misaligned pc"; return false; }
    // segment boundary: the next instruction and the pages it may bring in and write back must still fit.  An ordinary
    // instruction touches its own page and one more (in and out); an I/O ecall touches every page of its buffer, so its span
    // is priced before it runs (the word at pc is peeked without being charged to this segment yet).
    uint64_t worst = (uint64_t)lim.page_in_cycles * 2 + lim.page_out_cycles;
    bool io = false;
    {
      auto it = m.pages.find(m.pc >> PAGE_SHIFT);
      const uint32_t peek = it == m.pages.end() ? 0u : it->second[(m.pc & (PAGE_BYTES - 1)) >> 2];
      if (peek == 0x00000073u && (m.x[17] == 1 || m.x[17] == 2) && m.x[11] != 0) {
        const uint64_t bytes = m.x[17] == 1 ? (uint64_t)m.x[11] * 4 : (uint64_t)m.x[11];
        const uint64_t span = (((uint64_t)m.x[10] + bytes - 1) >> PAGE_SHIFT) - (m.x[10] >> PAGE_SHIFT) + 1;
        worst = lim.page_in_cycles + span * ((uint64_t)lim.page_in_cycles + (m.x[17] == 1 ? lim.page_out_cycles : 0));
        io = true;
      }
    }
    if (cur.info.user_cycles + 1 + paging_cycles() + worst > seg_budget) {
      if (cur.info.user_cycles != 0) {
        end_segment(2, 0);  // SystemSplit
        begin_segment();
      }
      if (1 + worst > seg_budget) {  // not even alone in a fresh segment
        err = io ? "an I/O ecall spans more pages than one segment can pay for: transfer in smaller pieces or raise segment_po2"
                 : "segment limit too small for a single instruction and its pages";
        return false;
      }
    }
    const uint32_t insn = load_word(m.pc);
    const uint32_t op = insn & 0x7f, rd = (insn >> 7) & 31, f3 = (insn >> 12) & 7, rs1 = (insn >> 15) & 31, rs2 = (insn >> 20) & 31, f7 = insn >> 25;
    const uint32_t a = m.x[rs1], b = m.x[rs2];
    const int32_t imm_i = (int32_t)insn >> 20;
    const int32_t imm_s = ((int32_t)(insn & 0xfe000000) >> 20) | (int32_t)((insn >> 7) & 31);
    const int32_t imm_b = ((int32_t)(insn & 0x80000000) >> 19) | (int32_t)((insn & 0x80) << 4) | (int32_t)((insn >> 20) & 0x7e0) | (int32_t)((insn >> 7) & 0x1e);
    const int32_t imm_j = ((int32_t)(insn & 0x80000000) >> 11) | (int32_t)(insn & 0xff000) | (int32_t)((insn >> 9) & 0x800) | (int32_t)((insn >> 20) & 0x7fe);
    uint32_t next = m.pc + 4, wr = 0;
    bool has_wr = false;
    r0h_preflight_row local;
    if (lim.keep_trace) {
      memset(&local, 0, sizeof local);
      local.cycle = cur.info.user_cycles; local.pc = m.pc; local.insn = insn; local.rs1_value = a; local.rs2_value = b;
      row = &local;
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
        const uint32_t fn = m.x[17], a0 = m.x[10], a1 = m.x[11];
        switch (fn) {
          case 0: *exit_kind = R0H_VM_HALTED; *exit_code = a0; running = false; break;
          case 4: *exit_kind = R0H_VM_PAUSED; *exit_code = a0; running = false; break;
          case 1:
            if (a0 & 3) { err = "READ_WORDS: misaligned destination"; return false; }
            if ((uint64_t)a1 * 4 > ((uint64_t)1 << 28)) { err = "READ_WORDS: count too large"; return false; }
            for (uint32_t i = 0; i < a1; i++) store_word(a0 + 4 * i, m.input_pos < m.input.size() ? m.input[m.input_pos++] : 0u);
            break;
          case 2:
            if ((uint64_t)a1 > ((uint64_t)1 << 28)) { err = "COMMIT: count too large"; return false; }
            for (uint32_t i = 0; i < a1; i++) {
              const uint32_t w = load_word((a0 + i) & ~3u);
              m.journal.push_back((uint8_t)(w >> (8 * ((a0 + i) & 3))));
            }
            break;
          case 3: m.x[10] = (uint32_t)m.cycles; if (row) { row->rd = 10; row->rd_after = m.x[10]; } break;
          default: err = "unknown ecall function"; return false;
        }
        break;
      }
      default: err = "illegal instruction"; return false;
    }
    if (has_wr && rd != 0) m.x[rd] = wr;
    if (row) {
      if (has_wr) { row->rd = rd; row->rd_after = rd ? wr : 0; }
      row->next_pc = next;
      cur.rows.push_back(local);
    }
    m.pc = next;
    m.cycles++;
    cur.info.user_cycles++;
    return running;
  }
};
}  // namespace

using namespace r0h;

extern "C" {

const char* r0h_vm_new(r0h_vm** out) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(out, "r0h_vm_new: NULL argument");
  *out = new r0h_vm;
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
  for (size_t i = 0; i < n; i++) {
    const uint32_t a = addr + 4 * (uint32_t)i;
    auto it = vm->pages.find(a >> PAGE_SHIFT);
    if (it == vm->pages.end()) it = vm->pages.emplace(a >> PAGE_SHIFT, std::vector<uint32_t>(PAGE_WORDS, 0)).first;
    it->second[(a & (PAGE_BYTES - 1)) >> 2] = words[i];
  }
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
    for (auto it = vm->pages.lower_bound((uint32_t)(z0 >> PAGE_SHIFT)); it != vm->pages.end() && ((uint64_t)it->first << PAGE_SHIFT) < z1; ++it) {
      const uint64_t base = (uint64_t)it->first << PAGE_SHIFT;
      for (uint32_t w = 0; w < PAGE_WORDS; w++)
        if (base + 4 * w >= z0 && base + 4 * w < z1) it->second[w] = 0;
    }
  }
  vm->pc = entry;
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
    auto it = vm->pages.find(a >> PAGE_SHIFT);
    words[i] = it == vm->pages.end() ? 0u : it->second[(a & (PAGE_BYTES - 1)) >> 2];
  }
  return nullptr;
}

const char* r0h_vm_run(r0h_vm* vm, const r0h_vm_limits* limits, int* exit_kind, uint32_t* exit_code) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(vm && limits && exit_kind && exit_code, "r0h_vm_run: NULL argument");
  R0H_REQUIRE(limits->segment_po2 >= 6 && limits->segment_po2 <= 24, "r0h_vm_run: segment_po2 %u outside [6, 24]", limits->segment_po2);
  R0H_REQUIRE(!vm->finished, "r0h_vm_run: this machine has already run to its end (one run per r0h_vm: load a new one)");
  vm->finished = true;
  Run run(*vm, *limits);
  run.begin_segment();
  *exit_kind = R0H_VM_LIMIT;
  *exit_code = 0;
  uint64_t executed = 0;
  bool running = true;
  while (running) {
    if (limits->max_cycles && executed >= limits->max_cycles) break;
    running = run.step(exit_kind, exit_code);
    if (run.err) {
      const uint32_t at = vm->pc;
      return make_error("guest trap at pc %#x after %llu cycles: %s", at, (unsigned long long)vm->cycles, run.err);
    }
    executed++;
  }
  // ExitCode of the last segment: Halted(code) / Paused(code) / SessionLimit
  if (*exit_kind == R0H_VM_HALTED) run.end_segment(0, *exit_code);
  else if (*exit_kind == R0H_VM_PAUSED) run.end_segment(1, *exit_code);
  else run.end_segment(2, 2);
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
// The DATA group of the trace circuit (tools/gen_circuit.py trace: R0H_TRACE_COLUMNS columns of 2^po2 rows, column-major,
// Montgomery words) from the preflight rows of segment i, and its three public inputs (first pc, pc after the last row, number of
// rows).  Row r of the witness is cycle r of the segment; rows past the end are blank.  32-bit words enter as 16-bit halves where
// the circuit only carries them, and reduced mod p where it does arithmetic on them (pc, addresses).
const char* r0h_vm_trace_witness(const r0h_vm* vm, size_t i, uint32_t po2, uint32_t* data_out, uint32_t globals_out[3]) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(vm && data_out && globals_out, "r0h_vm_trace_witness: NULL argument");
  R0H_REQUIRE(i < vm->segments.size(), "r0h_vm_trace_witness: segment %zu of %zu", i, vm->segments.size());
  const std::vector<r0h_preflight_row>& rows = vm->segments[i].rows;
  R0H_REQUIRE(!rows.empty(), "r0h_vm_trace_witness: segment %zu has no preflight rows (run with keep_trace)", i);
  R0H_REQUIRE(po2 <= R0H_MAX_PO2 && rows.size() <= ((size_t)1 << po2), "r0h_vm_trace_witness: %zu rows do not fit 2^%u", rows.size(), po2);
  const size_t n = (size_t)1 << po2;
  memset(data_out, 0, (size_t)R0H_TRACE_COLUMNS * n * 4);  // the Montgomery form of 0 is 0
  auto put = [&](uint32_t col, size_t r, uint32_t v) { data_out[(size_t)col * n + r] = enc(v % P); };
  for (size_t r = 0; r < rows.size(); r++) {
    const r0h_preflight_row& w = rows[r];
    put(0, r, 1);
    put(1, r, (uint32_t)r);
    put(2, r, w.pc);
    put(3, r, w.next_pc);
    put(4, r, w.next_pc == w.pc + 4 ? 1u : 0u);
    put(5, r, w.insn & 0xffffu); put(6, r, w.insn >> 16);
    put(7, r, w.rs1_value & 0xffffu); put(8, r, w.rs1_value >> 16);
    put(9, r, w.rs2_value & 0xffffu); put(10, r, w.rs2_value >> 16);
    put(11, r, w.rd);
    put(12, r, w.rd_after & 0xffffu); put(13, r, w.rd_after >> 16);
    put(14, r, w.mem_kind);
    put(15, r, w.mem_addr);
    put(16, r, w.mem_before & 0xffffu); put(17, r, w.mem_before >> 16);
    put(18, r, w.mem_after & 0xffffu); put(19, r, w.mem_after >> 16);
    for (uint32_t k = 0; k < 32; k++) put(20 + k, r, (w.insn >> k) & 1u);
  }
  // opcode classes that may leave the sequential path, pinned both ways by an inverse: flag = 1 iff the opcode is the class's
  // (blank rows past the end carry opcode 0 and therefore the inverses of -code)
  const uint32_t codes[3] = {0x6f, 0x67, 0x63};
  uint32_t inv_of[3][128];
  for (int c = 0; c < 3; c++)
    for (uint32_t op = 0; op < 128; op++) inv_of[c][op] = op == codes[c] ? 0u : inv(sub(enc(op), enc(codes[c])));
  for (size_t r = 0; r < n; r++) {
    const uint32_t op = r < rows.size() ? rows[r].insn & 0x7fu : 0u;
    for (int c = 0; c < 3; c++) {
      data_out[(size_t)(52 + c) * n + r] = op == codes[c] ? ONE : 0u;
      data_out[(size_t)(55 + c) * n + r] = inv_of[c][op];
    }
  }
  globals_out[0] = enc(rows.front().pc % P);
  globals_out[1] = enc(rows.back().next_pc % P);
  globals_out[2] = enc((uint32_t)(rows.size() % P));
  return nullptr;
  R0H_GUARD_END
}
const char* r0h_vm_journal(const r0h_vm* vm, const uint8_t** bytes, size_t* n) {
  R0H_REQUIRE(vm && bytes && n, "r0h_vm_journal: NULL argument");
  *bytes = vm->journal.data();
  *n = vm->journal.size();
  return nullptr;
}
// the receipt claim of segment i (csrc/claim.hip): pre/post system states and exit code from the run, the output digest on the last one
const char* r0h_vm_segment_claim(const r0h_vm* vm, size_t i, r0h_receipt_claim* out) {
  R0H_REQUIRE(vm && out, "r0h_vm_segment_claim: NULL argument");
  R0H_REQUIRE(i < vm->segments.size(), "r0h_vm_segment_claim: segment %zu of %zu", i, vm->segments.size());
  const r0h_vm_segment& s = vm->segments[i].info;
  memset(out, 0, sizeof *out);
  out->pre = s.pre;
  out->post = s.post;
  out->exit_system = s.exit_system;
  out->exit_user = s.exit_user;
  if (i + 1 == vm->segments.size() && s.exit_system <= 1) R0H_TRY(r0h_output_digest(vm->journal.data(), vm->journal.size(), nullptr, out->output_digest));
  return nullptr;
}

}  // extern "C"
