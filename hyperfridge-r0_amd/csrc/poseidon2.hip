// Poseidon2 (BabyBear, t=24, rate 16, x^7, 4+21+4 rounds) row hashing and Merkle folding for gfx950.
// Replaces risc0-zkp 3.0.4 hal `hash_rows` / `hash_fold` with the poseidon2 hash suite (CUDA side: risc0-sys 1.5.0
// poseidon2 kernels) -- SURVEY.md 8(a) a6-a8.
//
// One lane owns one row (hash_rows) or one parent node (hash_fold): the 24-word state lives in VGPRs, the round
// constants are wave-uniform and come in through scalar loads, and consecutive lanes read consecutive rows of each
// column so every column access of a wave is one contiguous 256-byte run.  This kernel is VALU-integer bound
// (about 1.36k Montgomery products per permutation), not HBM bound: see DESIGN.md for the arithmetic.
#include "poseidon2_device.hpp"

namespace r0h {

__global__ __launch_bounds__(256) void hash_rows_kernel(uint32_t* __restrict__ digests, const uint32_t* __restrict__ matrix,
                                                         uint32_t rows, uint32_t cols, const P2Consts* __restrict__ k) {
  uint32_t row = blockIdx.x * 256 + threadIdx.x;
  if (row >= rows) return;
  uint32_t c[P2_CELLS];
#pragma unroll
  for (int i = 0; i < P2_CELLS; i++) c[i] = 0;
  const uint32_t* src = matrix + row;
  uint32_t full = cols / P2_RATE, rem = cols % P2_RATE;
  // (prefetching the next rate block into registers was measured: no gain, the kernel is VALU-issue bound)
  for (uint32_t blk = 0; blk < full; blk++) {
#pragma unroll
    for (int i = 0; i < P2_RATE; i++) c[i] = src[(size_t)(blk * P2_RATE + i) * rows];
    p2_mix(c, k);
  }
  if (rem != 0 || cols == 0) {
#pragma unroll
    for (int i = 0; i < P2_RATE; i++) c[i] = (uint32_t)i < rem ? src[(size_t)(full * P2_RATE + i) * rows] : 0u;
    p2_mix(c, k);
  }
  uint4* dst = (uint4*)(digests + (size_t)row * 8);
  dst[0] = make_uint4(c[0], c[1], c[2], c[3]);
  dst[1] = make_uint4(c[4], c[5], c[6], c[7]);
}

// nodes[i] = H(nodes[2i] || nodes[2i+1]), output_size <= i < 2*output_size
__global__ __launch_bounds__(256) void hash_fold_kernel(uint32_t* __restrict__ nodes, uint32_t output_size,
                                                         const P2Consts* __restrict__ k) {
  uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= output_size) return;
  uint32_t i = output_size + t;
  const uint4* src = (const uint4*)(nodes + (size_t)2 * i * 8);
  uint32_t c[P2_CELLS];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    uint4 v = src[q];
    c[4 * q] = v.x; c[4 * q + 1] = v.y; c[4 * q + 2] = v.z; c[4 * q + 3] = v.w;
  }
#pragma unroll
  for (int q = 16; q < P2_CELLS; q++) c[q] = 0;
  p2_mix(c, k);
  uint4* dst = (uint4*)(nodes + (size_t)i * 8);
  dst[0] = make_uint4(c[0], c[1], c[2], c[3]);
  dst[1] = make_uint4(c[4], c[5], c[6], c[7]);
}

// Upper Merkle levels have fewer parents than the chip has lanes: one permutation per lane then takes a full
// single-wave latency (~26 us) per level, 17 levels per tree.  This variant spreads ONE permutation over 24 lanes of a
// 32-lane half-wave (one state cell per lane): S-boxes run in parallel, the external layer exchanges within quads and
// across the six quads by shuffles, the internal layer is a 5-step shuffle all-reduce.  Latency per level drops ~7x.
__device__ __forceinline__ uint32_t half_shfl(uint32_t v, uint32_t src_in_half) {
  return __shfl(v, (int)((threadIdx.x & 32u) | src_in_half), 64);
}
__device__ __forceinline__ uint32_t m_ext_lanes(uint32_t x, uint32_t l) {
  const uint32_t q = l & ~3u;
  uint32_t a = half_shfl(x, q), b = half_shfl(x, q | 1), d = half_shfl(x, q | 2), e = half_shfl(x, q | 3);
  uint32_t t0 = add(a, b), t1 = add(d, e);
  uint32_t t2 = add(add(b, b), t1), t3 = add(add(e, e), t0);
  uint32_t t1x2 = add(t1, t1), t0x2 = add(t0, t0);
  uint32_t t4 = add(add(t1x2, t1x2), t3), t5 = add(add(t0x2, t0x2), t2);
  uint32_t t6 = add(t3, t5), t7 = add(t2, t4);
  const uint32_t j = l & 3u;
  uint32_t y = j == 0 ? t6 : (j == 1 ? t5 : (j == 2 ? t7 : t4));
  // column sums over the six quads: pair neighbours, then add the two other pairs (lanes 24..31 carry don't-cares)
  uint32_t t = add(y, half_shfl(y, l ^ 4u));
  uint32_t s = add(t, add(half_shfl(t, (l + 8u) % 24u), half_shfl(t, (l + 16u) % 24u)));
  return add(y, s);
}
__global__ __launch_bounds__(256) void hash_fold_lanes_kernel(uint32_t* __restrict__ nodes, uint32_t output_size,
                                                               const P2Consts* __restrict__ k) {
  const uint32_t l = threadIdx.x & 31u, slot = (blockIdx.x * 256 + threadIdx.x) >> 5;  // one permutation per 32 lanes
  const bool live = slot < output_size, cell = l < 24u;
  const uint32_t node = output_size + (live ? slot : 0u), li = cell ? l : 0u;
  uint32_t c = (live && l < 16u) ? nodes[(size_t)2 * node * 8 + l] : 0u;
  const uint32_t dg = k->diag_canon[li], ds = k->diag_shoup[li];
  c = m_ext_lanes(c, l);
  for (int r = 0; r < P2_HALF_FULL; r++) c = m_ext_lanes(sbox7(add(c, k->rc_full[r][li])), l);
  for (int r = 0; r < P2_PARTIAL; r++) {
    if (l == 0) c = sbox7(add(c, k->rc_partial[r]));
    uint32_t sum = cell ? c : 0u;
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) sum = add(sum, half_shfl(sum, l ^ (uint32_t)off));
    c = add(sum, mul_const(c, dg, ds));
  }
  for (int r = P2_HALF_FULL; r < 2 * P2_HALF_FULL; r++) c = m_ext_lanes(sbox7(add(c, k->rc_full[r][li])), l);
  if (live && l < 8u) nodes[(size_t)node * 8 + l] = c;
}

}  // namespace r0h

using namespace r0h;

extern "C" {

const char* r0h_hash_rows(r0h_ctx* ctx, r0h_buf* digests, const r0h_buf* matrix, uint32_t rows, uint32_t cols) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(ctx && digests && matrix, "r0h_hash_rows: NULL argument");
  R0H_REQUIRE((size_t)rows * cols * 4 <= matrix->bytes, "r0h_hash_rows: %u x %u matrix exceeds the buffer", rows, cols);
  R0H_REQUIRE((size_t)rows * 32 <= digests->bytes, "r0h_hash_rows: %u digests exceed the output buffer", rows);
  R0H_REQUIRE(((uintptr_t)digests->ptr & 15) == 0, "r0h_hash_rows: digest buffer must be 16-byte aligned");
  if (!rows) return nullptr;
  KScope ks(ctx, "hash_rows_kernel", (double)rows * cols * 4 + (double)rows * 32);
  hipLaunchKernelGGL(hash_rows_kernel, dim3((rows + 255) / 256), dim3(256), 0, ctx->stream, u32(digests), u32(matrix), rows, cols, ctx->p2);
  hipError_t e = hipGetLastError();
  R0H_REQUIRE(e == hipSuccess, "hash_rows_kernel: %s", hipGetErrorString(e));
  return nullptr;
  R0H_GUARD_END
}

const char* r0h_hash_fold(r0h_ctx* ctx, r0h_buf* nodes, uint32_t output_size) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(ctx && nodes, "r0h_hash_fold: NULL argument");
  R0H_REQUIRE((size_t)output_size * 4 * 32 <= nodes->bytes, "r0h_hash_fold: output_size %u needs %zu bytes of nodes", output_size, (size_t)output_size * 128);
  R0H_REQUIRE(((uintptr_t)nodes->ptr & 15) == 0, "r0h_hash_fold: node buffer must be 16-byte aligned");
  if (!output_size) return nullptr;
  KScope ks(ctx, "hash_fold_kernel", (double)output_size * 96);
  if (output_size <= 8192) {  // fewer parents than the chip has SIMD slots (measured up to 8 K: 1.47 vs 1.52 ms per tree) (32x the instructions per permutation, ~7x less latency): spread each permutation over 24 lanes (latency, not throughput)
    hipLaunchKernelGGL(hash_fold_lanes_kernel, dim3((output_size * 32 + 255) / 256), dim3(256), 0, ctx->stream, u32(nodes), output_size, ctx->p2);
    hipError_t e = hipGetLastError();
    R0H_REQUIRE(e == hipSuccess, "hash_fold_lanes_kernel: %s", hipGetErrorString(e));
    return nullptr;
  }
  hipLaunchKernelGGL(hash_fold_kernel, dim3((output_size + 255) / 256), dim3(256), 0, ctx->stream, u32(nodes), output_size, ctx->p2);
  hipError_t e = hipGetLastError();
  R0H_REQUIRE(e == hipSuccess, "hash_fold_kernel: %s", hipGetErrorString(e));
  return nullptr;
  R0H_GUARD_END
}

// Hal::hash_fold(io, input_size, output_size): the trait names both sizes (the fold halves a level, so input_size == 2 * output_size)
const char* r0h_hash_fold_io(r0h_ctx* ctx, r0h_buf* io, uint32_t input_size, uint32_t output_size) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(input_size == 2 * (size_t)output_size, "r0h_hash_fold_io: a fold takes 2 * output_size = %zu digests in, not %u", 2 * (size_t)output_size, input_size);
  return r0h_hash_fold(ctx, io, output_size);
  R0H_GUARD_END
}

const char* r0h_merkle_build(r0h_ctx* ctx, r0h_buf* nodes, const r0h_buf* matrix, uint32_t rows, uint32_t cols) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(ctx && nodes && matrix, "r0h_merkle_build: NULL argument");
  R0H_REQUIRE(rows && (rows & (rows - 1)) == 0, "r0h_merkle_build: rows %u is not a power of two", rows);
  R0H_REQUIRE((size_t)rows * 2 * 32 <= nodes->bytes, "r0h_merkle_build: node buffer too small for %u rows", rows);
  r0h_buf leaves = *nodes;
  leaves.ptr = (char*)nodes->ptr + (size_t)rows * 32;
  leaves.bytes = (size_t)rows * 32;
  R0H_TRY(r0h_hash_rows(ctx, &leaves, matrix, rows, cols));
  for (uint32_t sz = rows / 2; sz >= 1; sz /= 2) R0H_TRY(r0h_hash_fold(ctx, nodes, sz));
  return nullptr;
  R0H_GUARD_END
}

}  // extern "C"
