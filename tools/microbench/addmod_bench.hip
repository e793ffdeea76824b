// Micro-benchmark: what a BabyBear modular addition really costs on gfx950, by formulation: the correction constant as a
// 32-bit literal (what the compiler emits for `x - P`), from an SGPR, and the three-input forms built on v_add3_u32 /
// v_min3_u32.  build: hipcc -O3 --offload-arch=gfx950 -o addmod_bench addmod_bench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

constexpr uint32_t P = 2013265921u;

struct K { uint32_t p, negp; };  // kernel arguments: uniform, live in SGPRs, opaque to constant folding

struct AddLiteral {  // x + y, then min(x, x - P) with P a literal
  static __device__ __forceinline__ uint32_t f(uint32_t a, uint32_t b, K) { uint32_t s = a + b, t = s - P; return t < s ? t : s; }
};
struct AddSgpr {  // the same with -P read from an SGPR
  static __device__ __forceinline__ uint32_t f(uint32_t a, uint32_t b, K k) { uint32_t s = a + b, t = s + k.negp; return t < s ? t : s; }
};
struct AddAdd3 {  // t = a + b - P in one v_add3_u32
  static __device__ __forceinline__ uint32_t f(uint32_t a, uint32_t b, K k) {
    uint32_t s = a + b, t;
    asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(t) : "v"(a), "v"(b), "s"(k.negp));
    return t < s ? t : s;
  }
};
struct Add3Min3 {  // three-input modular sum: y = a + b + (c - P) in (-p, 2p), then min3(y, y + P, y - P)
  static __device__ __forceinline__ uint32_t f(uint32_t a, uint32_t b, K k) {
    uint32_t c = a ^ 1u;  // a third operand below p
    uint32_t y, r;
    asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(y) : "v"(a), "v"(b), "s"(k.negp));
    y += c;
    uint32_t u = y + k.p, v = y + k.negp;
    asm volatile("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(y), "v"(u), "v"(v));
    return r;
  }
};
struct RawAdd { static __device__ __forceinline__ uint32_t f(uint32_t a, uint32_t b, K) { return a + (b ^ a); } };
struct RawMin { static __device__ __forceinline__ uint32_t f(uint32_t a, uint32_t b, K) { uint32_t s = a + b; return s < b ? s : b; } };
struct RawAddLit { static __device__ __forceinline__ uint32_t f(uint32_t a, uint32_t b, K) { return (a + 0x87ffffffu) ^ b; } };
struct RawAddSgpr { static __device__ __forceinline__ uint32_t f(uint32_t a, uint32_t b, K k) { return (a + k.negp) ^ b; } };

template <typename Op, int ILP>
__global__ __launch_bounds__(256) void chain(uint32_t* out, uint32_t seed, int iters, K k) {
  uint32_t x[ILP], y = (seed + threadIdx.x) % P;
#pragma unroll
  for (int i = 0; i < ILP; i++) x[i] = (seed * 2654435761u + i * 40503u + threadIdx.x + blockIdx.x * 977u) % P;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) x[i] = Op::f(x[i], y, k);
  }
  uint32_t acc = 0;
#pragma unroll
  for (int i = 0; i < ILP; i++) acc ^= x[i];
  if (acc == 0xdeadbeef) out[0] = acc;
}

template <typename Op>
static void run(const char* name, uint32_t* d, int instrs) {
  constexpr int ILP = 8;
  const int iters = 4096, blocks = 256 * 8;
  K k{P, 0u - P};
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  chain<Op, ILP><<<blocks, 256>>>(d, 1, 64, k);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 5; rep++) {
    hipEventRecord(a);
    chain<Op, ILP><<<blocks, 256>>>(d, 7 + rep, iters, k);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  double ops = (double)blocks * 256 * ILP * iters;
  // SIMD cycles per op: 1024 SIMDs, 64 lanes per wave instruction
  double cyc = best * 1e-3 * 2.37e9 * 1024 / (ops / 64);
  printf("%-12s %8.3f ms  %6.2f SIMD-cycles/op (%d VALU instructions written -> %.2f cycles each)\n", name, best, cyc, instrs, cyc / instrs);
}

int main() {
  uint32_t* d;
  hipMalloc(&d, 4096);
  run<RawAdd>("add+xor", d, 2);
  run<RawMin>("add+min", d, 2);
  run<RawAddLit>("addlit+xor", d, 2);
  run<RawAddSgpr>("addsgpr+xor", d, 2);
  run<AddLiteral>("mod_literal", d, 3);
  run<AddSgpr>("mod_sgpr", d, 3);
  run<AddAdd3>("mod_add3", d, 3);
  run<Add3Min3>("mod3_min3", d, 6);
  hipFree(d);
  return 0;
}
