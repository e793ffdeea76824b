// Micro-benchmark: throughput of BabyBear Montgomery-product formulations and of the raw integer instructions they
// are built from, on gfx950.  Used to choose the field kernel (DESIGN.md "Field arithmetic on the VALU").
// build: hipcc -O3 --offload-arch=gfx950 -o fpmul_bench fpmul_bench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

constexpr uint32_t P = 2013265921u, NPINV = 0x77ffffffu, PINV = 0x88000001u;

__device__ __forceinline__ uint32_t red1(uint32_t x) { uint32_t y = x - P; return y < x ? y : x; }

struct MulMad {  // 64-bit products (v_mad_u64_u32) + add-form reduction
  static __device__ __forceinline__ uint32_t f(uint32_t a, uint32_t b) {
    uint64_t t = (uint64_t)a * b;
    uint32_t m = (uint32_t)t * NPINV;
    uint64_t u = t + (uint64_t)m * P;
    return red1((uint32_t)(u >> 32));
  }
};
struct MulHiLo {  // explicit mul_lo / mul_hi, subtract-form reduction: hi(ab) - hi(m*P), m = lo(ab) * P^-1
  static __device__ __forceinline__ uint32_t f(uint32_t a, uint32_t b) {
    uint32_t lo = a * b, hi = __umulhi(a, b);
    uint32_t m = lo * PINV;
    uint32_t q = __umulhi(m, P);
    uint32_t r = hi - q;
    return hi < q ? r + P : r;
  }
};
struct MulF64 {  // double-double style: exact product through fma, Barrett quotient in floating point
  static __device__ __forceinline__ uint32_t f(uint32_t a, uint32_t b) {
    // NOT Montgomery: plain a*b mod p (would need canonical-form data); throughput probe only
    double da = (double)a, db = (double)b;
    double h = da * db, l = fma(da, db, -h);
    double q = floor(h * (1.0 / 2013265921.0));
    double r = fma(-q, 2013265921.0, h) + l;
    r = r < 0 ? r + 2013265921.0 : r;
    r = r >= 2013265921.0 ? r - 2013265921.0 : r;
    return (uint32_t)r;
  }
};
struct RawAdd { static __device__ __forceinline__ uint32_t f(uint32_t a, uint32_t b) { return a + (b ^ a); } };
struct RawMulLo { static __device__ __forceinline__ uint32_t f(uint32_t a, uint32_t b) { return a * b; } };
struct RawMulHi { static __device__ __forceinline__ uint32_t f(uint32_t a, uint32_t b) { return __umulhi(a, b); } };
struct RawMad64 {
  static __device__ __forceinline__ uint32_t f(uint32_t a, uint32_t b) {
    uint64_t t = (uint64_t)a * b + (((uint64_t)b << 32) | a);
    return (uint32_t)(t >> 32) ^ (uint32_t)t;
  }
};
struct RawMul24 { static __device__ __forceinline__ uint32_t f(uint32_t a, uint32_t b) { return __umul24(a, b) + b; } };
struct FpAdd { static __device__ __forceinline__ uint32_t f(uint32_t a, uint32_t b) { return red1(a + b); } };

template <typename Op, int ILP>
__global__ __launch_bounds__(256) void chain(uint32_t* out, uint32_t seed, int iters) {
  uint32_t x[ILP], y = seed + threadIdx.x;
#pragma unroll
  for (int k = 0; k < ILP; k++) x[k] = (seed * 2654435761u + k * 40503u + threadIdx.x + blockIdx.x * 977u) % P;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < ILP; k++) x[k] = Op::f(x[k], y);
  }
  uint32_t acc = 0;
#pragma unroll
  for (int k = 0; k < ILP; k++) acc ^= x[k];
  if (acc == 0xdeadbeef) out[0] = acc;  // keep the chain alive
}

template <typename Op>
static void run(const char* name, uint32_t* d) {
  constexpr int ILP = 8;
  const int iters = 4096, blocks = 256 * 8;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  chain<Op, ILP><<<blocks, 256>>>(d, 1, 64);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 5; rep++) {
    hipEventRecord(a);
    chain<Op, ILP><<<blocks, 256>>>(d, 7 + rep, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  double ops = (double)blocks * 256 * ILP * iters;
  printf("%-10s %8.3f ms  %8.2f Gop/s  (%.2f lane-ops/clk/CU at 2.4 GHz)\n", name, best, ops / best * 1e-6, ops / (best * 1e-3) / 256 / 2.4e9);
}

// Shader clock the chip sustains under this integer load: d(s_memtime) / d(s_memrealtime) x 100 MHz, one stamp pair
// around a long Montgomery-product loop per workgroup (MI355X_MICROARCH.md, DVFS give-back item 6).
__global__ __launch_bounds__(256) void clock_probe(uint32_t* sink, unsigned long long* stamps, int iters) {
  uint32_t x[8], y = 12345u + threadIdx.x;
#pragma unroll
  for (int k = 0; k < 8; k++) x[k] = (k * 40503u + threadIdx.x + blockIdx.x * 977u) % P;
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) x[k] = MulMad::f(x[k], y);
  }
  uint32_t acc = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) acc ^= x[k];
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (acc == 0xdeadbeef) sink[0] = acc;
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

static void probe_clock() {
  const int blocks = 256 * 8;
  uint32_t* sink; unsigned long long* d;
  hipMalloc(&sink, 64); hipMalloc(&d, blocks * 16);
  for (int rep = 0; rep < 20; rep++) clock_probe<<<blocks, 256>>>(sink, d, 200000);  // ~2 s of back-to-back launches
  hipDeviceSynchronize();
  unsigned long long h[blocks * 2];
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  double lo = 1e9, hi = 0, sum = 0;
  for (int b = 0; b < blocks; b++) {
    double ghz = (double)h[2 * b] / (double)h[2 * b + 1] * 0.1;
    lo = ghz < lo ? ghz : lo; hi = ghz > hi ? ghz : hi; sum += ghz;
  }
  printf("in-kernel shader clock under the Montgomery-product load: mean %.3f GHz (min %.3f, max %.3f over %d workgroups)\n", sum / blocks, lo, hi, blocks);
  hipFree(sink); hipFree(d);
}

int main() {
  uint32_t* d;
  hipMalloc(&d, 4096);
  run<RawAdd>("add+xor", d);
  run<FpAdd>("fp_add", d);
  run<RawMulLo>("mul_lo", d);
  run<RawMulHi>("mul_hi", d);
  run<RawMad64>("mad_u64", d);
  run<RawMul24>("mul_u24", d);
  run<MulMad>("mont_mad", d);
  run<MulHiLo>("mont_hilo", d);
  run<MulF64>("mod_f64", d);
  probe_clock();
  hipFree(d);
  return 0;
}
