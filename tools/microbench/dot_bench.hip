// Micro-benchmark: cost per term of a lazily accumulated dot product sum_i u_i * k_i mod p (64-bit accumulator, the high word
// brought below p every second term so the sum never overflows) against the per-term cost of the Shoup form used in the
// Poseidon2 partial rounds.  build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I include -I hyperfridge-r0_amd/csrc ...
#include "internal.hpp"

#include <stdio.h>

using namespace r0h;

// acc < 2^64 invariant: after a fix hi < p, so acc < p 2^32; two more products add < 2 p^2: total < 2^64
__device__ __forceinline__ void fix_hi(uint64_t& acc) {
  uint32_t hi = (uint32_t)(acc >> 32);
  hi = reduce1(reduce1(hi));  // hi < 2^32 < 3p: at most two subtractions
  acc = ((uint64_t)hi << 32) | (uint32_t)acc;
}
__device__ __forceinline__ void fix_hi1(uint64_t& acc) {  // when hi < 2p is known
  uint32_t hi = reduce1((uint32_t)(acc >> 32));
  acc = ((uint64_t)hi << 32) | (uint32_t)acc;
}

template <int MODE, int TERMS, int NACC>
__global__ __launch_bounds__(256) void dot_kernel(uint32_t* out, const uint32_t* __restrict__ consts, int iters) {
  uint32_t u[TERMS];
#pragma unroll
  for (int i = 0; i < TERMS; i++) u[i] = (threadIdx.x * 2654435761u + blockIdx.x * 977u + i * 40503u) % P;
  uint32_t res[NACC];
#pragma unroll
  for (int a = 0; a < NACC; a++) res[a] = a;
#pragma unroll 1
  for (int it = 0; it < iters; it++) {
    const uint32_t* k = consts + (it & 15) * TERMS;  // wave-uniform: scalar loads
#pragma unroll
    for (int a = 0; a < NACC; a++) {
      if (MODE == 0) {  // lazy dot: products 1..4 free, then a fix every second product
        uint64_t acc = (uint64_t)res[a] * k[0];
#pragma unroll
        for (int i = 1; i < TERMS; i++) {
          if (i >= 4 && (i & 1) == 0) fix_hi1(acc);  // acc < p 2^32 + 2 p^2 -> hi < p + 2p^2/2^32 < 2p
          acc += (uint64_t)(u[i] ^ a) * k[i];
        }
        res[a] = reduce64(acc < ((uint64_t)P << 33) ? acc : acc - ((uint64_t)P << 33));
      } else {  // Shoup form: every product reduced, then summed by modular adds
        uint32_t s = res[a];
#pragma unroll
        for (int i = 1; i < TERMS; i++) s = add(s, mul_const(u[i] ^ a, k[i], k[i] + 1));
        res[a] = s;
      }
    }
  }
  uint32_t acc = 0;
#pragma unroll
  for (int a = 0; a < NACC; a++) acc ^= res[a];
  if (acc == 0xdeadbeef) out[0] = acc;
}

template <int MODE, int TERMS, int NACC>
static void run(const char* name, uint32_t* d, const uint32_t* dk, int iters, int waves) {
  const int blocks = 256 * waves;
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  dot_kernel<MODE, TERMS, NACC><<<blocks, 256>>>(d, dk, 8);
  (void)hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 3; rep++) {
    (void)hipEventRecord(a);
    dot_kernel<MODE, TERMS, NACC><<<blocks, 256>>>(d, dk, iters);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  double cyc = best * 1e-3 * 2.37e9 / ((double)iters * waves * NACC * (TERMS - 1));
  printf("%-22s terms %2d  accumulators %d  waves/SIMD %d  %8.3f ms  %6.2f SIMD-cycles per term\n", name, TERMS, NACC, waves, best, cyc);
}

int main() {
  uint32_t *d, *dk;
  (void)hipMalloc(&d, 4096);
  (void)hipMalloc(&dk, 16 * 32 * 4);
  uint32_t h[16 * 32];
  for (int i = 0; i < 16 * 32; i++) h[i] = (uint32_t)(((uint64_t)i * 2654435761u + 12345u) % P);
  (void)hipMemcpy(dk, h, sizeof h, hipMemcpyHostToDevice);
  for (int w = 2; w <= 3; w++) {
    run<0, 24, 1>("lazy dot", d, dk, 4000, w);
    run<0, 24, 3>("lazy dot", d, dk, 2000, w);
    run<1, 24, 1>("shoup + modular adds", d, dk, 4000, w);
    run<1, 24, 3>("shoup + modular adds", d, dk, 2000, w);
  }
  return 0;
}
