// The image proof: what lets `receipt.verify(image_id)` (verifier/src/main.rs:124-126; risc0-zkvm 3.0.5 `Receipt::verify`) check a
// trace-circuit session WITHOUT the program image in hand.  r0h_receipt_verify_elf completes the session's memory argument by adding
// 1 / fingerprint(address, word) over the ELF's own words; a verifier with the 32 bytes of the image id cannot.  The image circuit
// (tools/image_circuit.py, circuits/image.r0c) proves that sum for it: a Poseidon2 sponge over the image's word list, computed inside
// the proof, gives the digest the image id names as the root of the initial memory state (csrc/rv32im.cpp initial_root), and the same
// rows add every word's fraction under the session's challenge to a running sum whose total is a public input.  The verifier checks
// the seal, takes the digest from the first claim's pre-state (which it holds against the image id) and the total for its balance.
//
// Host: r0h_image_witness (the circuit's DATA group for an ELF: sponge rows + the four validity flags), r0h_image_po2.  Device:
// r0h_prove_image (witness uploaded, total accumulated, proved like any segment; a few milliseconds for a 13 KiB image).
#include <string.h>

#include <algorithm>
#include <memory>
#include <vector>

#include "../../include/r0hip_circuit.h"
#include "circuit.hpp"
#include "receipt_types.hpp"

using namespace r0h;

namespace r0h {
bool is_image_circuit(const r0h_circuit& c) {
  return !memcmp(c.info, "R0HIP_IMAGE:v1__", 16) && c.n_global == R0H_IMAGE_GLOBALS && c.n_late == R0H_IMAGE_LATE_GLOBALS && c.has_sponge &&
         c.group_size[R0H_GROUP_DATA] == R0H_IMAGE_COLUMNS && c.sponge_data == 0 && c.sponge_global == 0;
}
// the DATA group for an image: [R0H_IMAGE_COLUMNS][2^po2] Montgomery words, column-major
const char* image_witness(const std::vector<std::pair<uint32_t, uint32_t>>& image, uint32_t po2, uint32_t* data, uint32_t digest[8]) {
  std::vector<uint32_t> stream;
  image_stream(image, stream);
  const size_t n = (size_t)1 << po2, n_blocks = stream.size() / 16;
  R0H_REQUIRE(n_blocks * R0H_SPONGE_PERIOD < n, "the image's %zu words take %zu rows: the image trace has 2^%u", image.size(), n_blocks * R0H_SPONGE_PERIOD, po2);
  memset(data, 0, (size_t)R0H_IMAGE_COLUMNS * n * 4);
  std::unique_ptr<P2Consts> k(new P2Consts);
  p2_default_host(*k);
  size_t used = 0;
  p2_sponge_rows_host(*k, stream.data(), stream.size(), data, n, &used);
  for (size_t q = 0; q < n_blocks; q++) {  // the flags: the mask's bits, on the row that absorbs the block
    const size_t there = std::min<size_t>(4, image.size() - std::min<size_t>(image.size(), 4 * q));
    for (size_t j = 0; j < there; j++) data[(R0H_SPONGE_DATA_COLUMNS + j) * n + q * R0H_SPONGE_PERIOD] = ONE;
  }
  if (digest) p2_hash_elems_host(*k, stream.data(), stream.size(), digest);
  return nullptr;
}
}  // namespace r0h

extern "C" {

const char* r0h_image_po2(const uint8_t* elf, size_t elf_len, uint32_t* po2_out) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(elf && po2_out, "r0h_image_po2: NULL argument");
  std::vector<std::pair<uint32_t, uint32_t>> image;
  uint32_t entry;
  uint8_t id[32];
  R0H_TRY(elf_image(elf, elf_len, image, &entry, id));
  const size_t rows = (image.empty() ? 1 : (image.size() + 3) / 4) * R0H_SPONGE_PERIOD;
  uint32_t po2 = 9;  // the prover's smallest trace
  while (((size_t)1 << po2) <= rows) po2++;
  R0H_REQUIRE(po2 <= R0H_MAX_PO2, "r0h_image_po2: an image of %zu words does not fit a trace", image.size());
  *po2_out = po2;
  return nullptr;
  R0H_GUARD_END
}

const char* r0h_image_witness(const uint8_t* elf, size_t elf_len, uint32_t po2, uint32_t* data_out, uint32_t globals_out[R0H_IMAGE_GLOBALS]) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(elf && data_out && globals_out, "r0h_image_witness: NULL argument");
  R0H_REQUIRE(po2 >= 6 && po2 <= R0H_MAX_PO2, "r0h_image_witness: po2 %u outside [6, %u]", po2, R0H_MAX_PO2);
  std::vector<std::pair<uint32_t, uint32_t>> image;
  uint32_t entry;
  uint8_t id[32];
  R0H_TRY(elf_image(elf, elf_len, image, &entry, id));
  memset(globals_out, 0, R0H_IMAGE_GLOBALS * 4);
  return image_witness(image, po2, data_out, globals_out);
  R0H_GUARD_END
}

// Sessions begun on `ctx` afterwards attach an image proof to their receipts (NULL: stop).  The circuit must be loaded on this context.
const char* r0h_ctx_set_image_circuit(r0h_ctx* ctx, const r0h_circuit* image_circuit) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(ctx, "r0h_ctx_set_image_circuit: ctx is NULL");
  if (image_circuit) {
    R0H_REQUIRE(is_image_circuit(*image_circuit), "r0h_ctx_set_image_circuit: the circuit is not the image circuit (circuits/image.r0c) this library was built for");
    R0H_REQUIRE(image_circuit->ctx == ctx, "r0h_ctx_set_image_circuit: the circuit was loaded on another context");
  }
  ctx->image_circuit = image_circuit;
  return nullptr;
  R0H_GUARD_END
}

// One image proof on `ctx`: the seal's public inputs are the image's digest (8), the challenge given (16), the total (4).
const char* r0h_prove_image(r0h_ctx* ctx, const r0h_circuit* c, const uint8_t* elf, size_t elf_len, const uint32_t challenge[16], uint32_t* seal_out,
                            size_t seal_capacity_words, size_t* seal_words_out) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(ctx && c && elf && challenge && seal_words_out, "r0h_prove_image: NULL argument");
  R0H_REQUIRE(is_image_circuit(*c), "r0h_prove_image: the circuit is not the image circuit (circuits/image.r0c) this library was built for");
  for (int i = 0; i < 16; i++) R0H_REQUIRE(challenge[i] < P, "r0h_prove_image: challenge word %d is not a canonical field word", i);
  R0H_TRY_HIP(hipSetDevice(ctx->device));
  std::vector<std::pair<uint32_t, uint32_t>> image;
  uint32_t entry, po2 = 9;
  uint8_t id[32];
  R0H_TRY(elf_image(elf, elf_len, image, &entry, id));
  std::vector<uint32_t> stream, global(R0H_IMAGE_GLOBALS, 0);
  image_stream(image, stream);
  const size_t n_blocks = stream.size() / 16, rows = n_blocks * R0H_SPONGE_PERIOD;
  while (((size_t)1 << po2) <= rows) po2++;
  R0H_REQUIRE(po2 <= R0H_MAX_PO2, "r0h_prove_image: an image of %zu words does not fit a trace", image.size());
  const size_t n = (size_t)1 << po2;
  {
    std::unique_ptr<P2Consts> k(new P2Consts);
    p2_default_host(*k);
    p2_hash_elems_host(*k, stream.data(), stream.size(), global.data());
  }
  memcpy(&global[R0H_IMAGE_GAMMA], challenge, 64);
  r0h_buf *code = nullptr, *data = nullptr, *scratch = nullptr;
  struct Free { r0h_buf*& b; ~Free() { if (b) r0h_buf_free(b); } } f0{code}, f1{data}, f2{scratch};
  const size_t data_bytes = (size_t)R0H_IMAGE_COLUMNS * n * 4;
  R0H_TRY(buf_alloc_pooled(ctx, (size_t)c->group_size[R0H_GROUP_CODE] * n * 4, &code));
  R0H_TRY(buf_alloc_pooled(ctx, data_bytes, &data));
  R0H_TRY(buf_alloc_pooled(ctx, data_bytes, &scratch));
  R0H_TRY(r0h_witgen(ctx, c, po2, 0, code, scratch, nullptr));  // (the CODE columns; the DATA it fills beside them is not the image's)
  // the witness: the sponge's rows over the blocks (only the rows in use travel), the flags on the rows that absorb a block
  R0H_TRY(sponge_plant(ctx, c, po2, stream.data(), stream.size(), data));
  std::vector<uint32_t> flags(4 * n_blocks, 0u);  // [flag][block]
  for (size_t k = 0; k < image.size(); k++) flags[(k % 4) * n_blocks + k / 4] = ONE;
  uint32_t* first_flag = (uint32_t*)data->ptr + (size_t)R0H_SPONGE_DATA_COLUMNS * n;
  R0H_TRY_HIP(hipMemsetAsync(first_flag, 0, 4 * n * 4, ctx->stream));
  for (size_t j = 0; j < 4; j++)  // one word every thirty rows
    R0H_TRY_HIP(hipMemcpy2DAsync(first_flag + j * n, (size_t)R0H_SPONGE_PERIOD * 4, flags.data() + j * n_blocks, 4, 4, n_blocks, hipMemcpyHostToDevice, ctx->stream));
  R0H_TRY(r0h_logup_totals(ctx, c, po2, code, data, global.data()));
  return r0h_prove_segment(ctx, c, po2, code, data, global.data(), seal_out, seal_capacity_words, seal_words_out);
  R0H_GUARD_END
}

}  // extern "C"
