// Receipt / claim types shared by receipt.cpp (JSON reader and writer) and claim.cpp (digests, receipt verification).
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/r0hip.h"
#include "circuit.hpp"

struct r0h_receipt {
  int kind = R0H_RECEIPT_FAKE;
  std::vector<uint8_t> journal;
  struct Segment {
    std::vector<uint32_t> seal;
    uint32_t index = 0;
    std::string hashfn;
    bool has_claim = false;
    r0h_receipt_claim claim{};
    uint8_t verifier_parameters[32] = {0};
  };
  std::vector<Segment> segments;
  std::vector<uint32_t> image_seal;  // optional: the image proof of a trace-circuit session (csrc/image.cpp), `inner.Composite.image_proof.seal`
  bool has_metadata = false;  // risc0 >= 1.0 `Receipt.metadata`; the reference's (older) Fake fixtures have none
  uint8_t verifier_parameters[32] = {0};
};

namespace r0h {
struct Sha256 {
  uint32_t h[8];
  uint64_t total;
  uint8_t buf[64];
  size_t fill;
  Sha256() { reset(); }
  void reset();
  void update(const void* data, size_t n);
  void finish(uint8_t out[32]);

 private:
  void block(const uint8_t* p);
};
void sha256(const void* data, size_t n, uint8_t out[32]);
void tagged_struct(const char* tag, const uint8_t (*down)[32], size_t n_down, const uint32_t* data, size_t n_data, uint8_t out[32]);
void system_state_digest(const r0h_system_state& st, uint8_t out[32]);
void claim_digest(const r0h_receipt_claim& c, uint8_t out[32]);
void claim_globals(const uint8_t digest[32], uint32_t out[8]);
bool is_trace_circuit(const r0h_circuit& circ);
bool is_image_circuit(const r0h_circuit& circ);  // image.cpp
bool trace_seal_carries_claim(const uint32_t* seal, const r0h_receipt_claim& claim);
void session_challenge(const uint32_t* records, size_t n_records, uint32_t out[16]);
void image_stream(const std::vector<std::pair<uint32_t, uint32_t>>& image, std::vector<uint32_t>& words_out);  // rv32im.cpp: the image circuit's sponge blocks (Montgomery)
void image_digest(const std::vector<std::pair<uint32_t, uint32_t>>& image, uint32_t digest_out[8]);             // ... and their digest: the root of the initial memory state
const char* elf_image(const uint8_t* elf, size_t n, std::vector<std::pair<uint32_t, uint32_t>>& image, uint32_t* entry, uint8_t image_id[32]);  // rv32im.cpp
}  // namespace r0h
