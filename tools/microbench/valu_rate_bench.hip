// Micro-benchmark: issue cost of single gfx950 integer VALU instructions (inline asm, so the instruction is exactly the one
// named), as SIMD cycles per wave-instruction with 8 waves per SIMD and 8 independent chains per wave.
// build: hipcc -O3 --offload-arch=gfx950 -o valu_rate_bench valu_rate_bench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define OP2(NAME, TEXT)                                                                       \
  struct NAME {                                                                               \
    static __device__ __forceinline__ uint32_t f(uint32_t a, uint32_t b, uint32_t s) {        \
      uint32_t r;                                                                             \
      asm volatile(TEXT : "=v"(r) : "v"(a), "v"(b), "s"(s));                                  \
      return r;                                                                               \
    }                                                                                         \
  };
OP2(Add, "v_add_u32 %0, %1, %2")
OP2(AddS, "v_add_u32 %0, %3, %1")
OP2(Sub, "v_sub_u32 %0, %1, %2")
OP2(Xor, "v_xor_b32 %0, %1, %2")
OP2(MinU, "v_min_u32 %0, %1, %2")
OP2(MaxU, "v_max_u32 %0, %1, %2")
OP2(MinI, "v_min_i32 %0, %1, %2")
OP2(Min3, "v_min3_u32 %0, %1, %2, %1")
OP2(Add3, "v_add3_u32 %0, %1, %2, %3")
OP2(LshlAdd, "v_lshl_add_u32 %0, %1, 1, %2")
OP2(AddLshl, "v_add_lshl_u32 %0, %1, %2, 1")
OP2(AndOr, "v_and_or_b32 %0, %1, %2, %3")
OP2(Bfi, "v_bfi_b32 %0, %1, %2, %3")
OP2(Ashr, "v_ashrrev_i32 %0, 31, %1")
OP2(MulLo, "v_mul_lo_u32 %0, %1, %2")
OP2(MulHi, "v_mul_hi_u32 %0, %1, %2")
OP2(MulU24, "v_mul_u32_u24 %0, %1, %2")
OP2(MadU24, "v_mad_u32_u24 %0, %1, %2, %1")
OP2(SubCoCnd, "v_sub_co_u32 %0, vcc, %1, %2\n v_cndmask_b32 %0, %0, %1, vcc")
OP2(CmpCnd, "v_cmp_lt_u32 vcc, %1, %2\n v_cndmask_b32 %0, %2, %1, vcc")
OP2(AddcPair, "v_add_co_u32 %0, vcc, %1, %2\n v_addc_co_u32 %0, vcc, %0, %2, vcc")

struct LshlAdd64 {  // 64-bit add in one instruction
  static __device__ __forceinline__ uint32_t f(uint32_t a, uint32_t b, uint32_t) {
    uint64_t r, x = ((uint64_t)b << 32) | a, y = ((uint64_t)a << 32) | b;
    asm volatile("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(r) : "v"(x), "v"(y));
    return (uint32_t)(r >> 32) ^ (uint32_t)r;
  }
};
struct Lshl64 {
  static __device__ __forceinline__ uint32_t f(uint32_t a, uint32_t b, uint32_t) {
    uint64_t r, x = ((uint64_t)b << 32) | a;
    asm volatile("v_lshlrev_b64 %0, 1, %1" : "=v"(r) : "v"(x));
    return (uint32_t)(r >> 32) ^ (uint32_t)r;
  }
};
struct Mad64Zero {  // zero addend
  static __device__ __forceinline__ uint32_t f(uint32_t a, uint32_t b, uint32_t) {
    uint64_t r;
    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b) : "vcc");
    return (uint32_t)(r >> 32) ^ (uint32_t)r;
  }
};
struct Mad64 {
  static __device__ __forceinline__ uint32_t f(uint32_t a, uint32_t b, uint32_t) {
    uint64_t r, c = ((uint64_t)b << 32) | a;
    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c) : "vcc");
    return (uint32_t)(r >> 32);
  }
};

template <typename Op, int ILP>
__global__ __launch_bounds__(256) void chain(uint32_t* out, uint32_t seed, int iters, uint32_t s) {
  uint32_t x[ILP], y = seed + threadIdx.x;
#pragma unroll
  for (int i = 0; i < ILP; i++) x[i] = seed * 2654435761u + i * 40503u + threadIdx.x + blockIdx.x * 977u;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) x[i] = Op::f(x[i], y, s);
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
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  chain<Op, ILP><<<blocks, 256>>>(d, 1, 64, 0x87ffffffu);
  (void)hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 5; rep++) {
    (void)hipEventRecord(a);
    chain<Op, ILP><<<blocks, 256>>>(d, 7 + rep, iters, 0x87ffffffu);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  double ops = (double)blocks * 256 * ILP * iters;
  double cyc = best * 1e-3 * 2.37e9 * 1024 / (ops / 64);
  printf("%-14s %8.3f ms  %6.2f SIMD-cycles per wave-op  (%d instruction%s -> %.2f each)\n", name, best, cyc, instrs, instrs > 1 ? "s" : "", cyc / instrs);
}

int main() {
  uint32_t* d;
  (void)hipMalloc(&d, 4096);
  run<Add>("v_add_u32", d, 1);
  run<AddS>("v_add_u32 sgpr", d, 1);
  run<Sub>("v_sub_u32", d, 1);
  run<Xor>("v_xor_b32", d, 1);
  run<MinU>("v_min_u32", d, 1);
  run<MaxU>("v_max_u32", d, 1);
  run<MinI>("v_min_i32", d, 1);
  run<Min3>("v_min3_u32", d, 1);
  run<Add3>("v_add3_u32", d, 1);
  run<LshlAdd>("v_lshl_add_u32", d, 1);
  run<AddLshl>("v_add_lshl_u32", d, 1);
  run<AndOr>("v_and_or_b32", d, 1);
  run<Bfi>("v_bfi_b32", d, 1);
  run<Ashr>("v_ashrrev_i32", d, 1);
  run<MulLo>("v_mul_lo_u32", d, 1);
  run<MulHi>("v_mul_hi_u32", d, 1);
  run<MulU24>("v_mul_u32_u24", d, 1);
  run<MadU24>("v_mad_u32_u24", d, 1);
  run<Mad64>("v_mad_u64_u32", d, 1);
  run<Mad64Zero>("v_mad_u64 (+0) +xor", d, 2);
  run<LshlAdd64>("v_lshl_add_u64 +xor", d, 2);
  run<Lshl64>("v_lshlrev_b64 +xor", d, 2);
  run<SubCoCnd>("sub_co+cndmask", d, 2);
  run<CmpCnd>("cmp+cndmask", d, 2);
  run<AddcPair>("add_co+addc", d, 2);
  (void)hipFree(d);
  return 0;
}
