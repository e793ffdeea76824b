// Micro-benchmark: HBM rate of the strided NTT pass's access pattern -- a block moves a [2^10][T] tile of a column-major
// 2^22-word column (rows 2^12 words apart) through registers -- for T = 16 words (64-byte rows, what ntt_strided16 does) and
// T = 32 (128-byte rows).  build: hipcc -O3 --offload-arch=gfx950 -o tile_copy_bench tile_copy_bench.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <int WORDS>  // words per thread per row: 1 -> T = 16, 2 -> T = 32
__global__ __launch_bounds__(1024) void tile_copy(uint32_t* io, uint32_t n, uint32_t L) {
  const uint32_t t = threadIdx.x & 15, q = threadIdx.x >> 4, lo = (blockIdx.x * 16 + t) * WORDS;
  uint32_t* col = io + ((size_t)blockIdx.y << n);
  uint32_t x[16][WORDS];
#pragma unroll
  for (int j = 0; j < 16; j++) {
    const uint32_t* p = col + ((size_t)(q * 16 + j) << L) + lo;
    if (WORDS == 1) x[j][0] = p[0];
    else { uint2 v = *(const uint2*)p; x[j][0] = v.x; x[j][WORDS - 1] = v.y; }
  }
#pragma unroll
  for (int j = 0; j < 16; j++) {
    uint32_t* p = col + ((size_t)(q * 16 + j) << L) + lo;
    if (WORDS == 1) p[0] = x[j][0] + 1u;
    else *(uint2*)p = make_uint2(x[j][0] + 1u, x[j][WORDS - 1] + 1u);
  }
}

template <int WORDS>
static void run(uint32_t* d, int cols) {
  const uint32_t n = 22, L = 12;
  dim3 grid((1u << L) / (16 * WORDS), cols);
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  tile_copy<WORDS><<<grid, 1024>>>(d, n, L);
  (void)hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 5; rep++) {
    (void)hipEventRecord(a);
    tile_copy<WORDS><<<grid, 1024>>>(d, n, L);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  double bytes = 2.0 * cols * (double)(1u << n) * 4;
  printf("rows of %3d bytes: %7.3f ms for %d columns  %7.1f GB/s (read + write)\n", 64 * WORDS, best, cols, bytes / best * 1e-6);
}

int main() {
  const int cols = 192;
  uint32_t* d;
  (void)hipMalloc(&d, (size_t)cols << 24);
  (void)hipMemset(d, 0, (size_t)cols << 24);
  run<1>(d, cols);
  run<2>(d, cols);
  return 0;
}
