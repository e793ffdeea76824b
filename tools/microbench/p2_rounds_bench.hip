// Micro-benchmark: SIMD cycles one wave spends per Poseidon2 full round / partial round / whole permutation, as a function
// of how many waves share a SIMD -- the numbers DESIGN.md's VALU budget for hash_rows is checked against.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I include -I hyperfridge-r0_amd/csrc -o p2_rounds_bench p2_rounds_bench.hip \
//        -L hyperfridge-r0_amd -lr0hip -Wl,-rpath,$PWD/hyperfridge-r0_amd
#include "poseidon2_device.hpp"

#include <stdio.h>


using namespace r0h;

template <int MODE>
__global__ __launch_bounds__(256) void rounds_kernel(uint32_t* out, const P2Consts* __restrict__ k, int iters) {
  uint32_t c[P2_CELLS];
#pragma unroll
  for (int i = 0; i < P2_CELLS; i++) c[i] = (threadIdx.x * 2654435761u + blockIdx.x * 977u + i * 40503u) % P;
  uint32_t rest = sum_lanes_1_to_23(c);
#pragma unroll 1
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) p2_full_round(c, k->rc_full[it & 7]);
    if (MODE == 1) p2_partial_rounds(c, k);
    if (MODE == 2) p2_mix(c, k);
    if (MODE == 3) m_ext(c);
    if (MODE == 4) {
#pragma unroll
      for (int i = 0; i < P2_CELLS; i++) c[i] = sbox7(c[i]);
    }
  }
  uint32_t acc = rest;
#pragma unroll
  for (int i = 0; i < P2_CELLS; i++) acc ^= c[i];
  if (acc == 0xdeadbeef) out[0] = acc;
}

template <int MODE>
static void run(const char* name, uint32_t* d, const P2Consts* dk, int iters, int waves_per_simd) {
  const int blocks = 256 * waves_per_simd;  // 256 threads = one wave on each SIMD of a CU; 256 CUs
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  rounds_kernel<MODE><<<blocks, 256>>>(d, dk, 8);
  (void)hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 3; rep++) {
    (void)hipEventRecord(a);
    rounds_kernel<MODE><<<blocks, 256>>>(d, dk, iters);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  // every SIMD runs waves_per_simd waves concurrently; cycles of SIMD time per (wave, iteration)
  double cyc = best * 1e-3 * 2.37e9 / ((double)iters * waves_per_simd);
  printf("%-14s waves/SIMD %d  %8.3f ms  %9.1f SIMD-cycles per wave-iteration\n", name, waves_per_simd, best, cyc);
}

int main() {
  uint32_t* d;
  P2Consts hk, *dk;
  p2_default_host(hk);  // from libr0hip.so
  (void)hipMalloc(&d, 4096);
  (void)hipMalloc(&dk, sizeof hk);
  (void)hipMemcpy(dk, &hk, sizeof hk, hipMemcpyHostToDevice);
  for (int w = 1; w <= 5; w++) {
    run<0>("full round", d, dk, 4000, w);
    run<1>("21 partial", d, dk, 400, w);
    run<3>("m_ext", d, dk, 8000, w);
    run<4>("24 x sbox7", d, dk, 4000, w);
    run<2>("permutation", d, dk, 200, w);
  }
  return 0;
}
