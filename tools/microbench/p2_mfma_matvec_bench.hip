// Micro-benchmark: the constant-matrix part of Poseidon2's deferred partial rounds -- sigma-pre_r = sum_i D_i^r u_i
// for r = 1..20 over the 23 entry lanes of a row, 460 of the 1,176 multiply-add terms of the permutation and independent of the
// round recurrence -- computed two ways, same canonical words:
//   VALU   20 lazily accumulated dot products of 23 terms (what poseidon2_device.hpp does today: v_mad_u64_u32 per term)
//   MFMA   signed 8-bit limbs of the 23 lanes (one add + one xor per word) through LDS into v_mfma_i32_16x16x64_i8 against
//          limb-expanded constant tables: 7 partial sums S_s = sum_{a+b=s} u_a d_b per output (rows on M, rounds on N, 28 MFMAs per
//          16 rows), recombined as sum_s S_s 2^(8s) mod p by 7 v_mad_i64_i32 + one Montgomery reduction in the C layout, transposed
//          back through LDS.
// Prints SIMD cycles per wave-iteration for both at 1..3 waves per SIMD and checks that the two agree word for word.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I include -I hyperfridge-r0_amd/csrc -o p2_mfma_matvec_bench p2_mfma_matvec_bench.hip \
//        -L hyperfridge-r0_amd -lr0hip -Wl,-rpath,$PWD/hyperfridge-r0_amd
#include "poseidon2_device.hpp"

#include <stdio.h>

#include <vector>

using namespace r0h;

constexpr int NU = 23, NR = 20;
typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void seed_lanes(uint32_t (&u)[NU]) {
#pragma unroll
  for (int i = 0; i < NU; i++) u[i] = (uint32_t)(((uint64_t)(threadIdx.x * 2654435761u + blockIdx.x * 977u + i * 40503u + 12345u) * 7919u) % P);
}
__device__ __forceinline__ void feed_back(uint32_t (&u)[NU], const uint32_t (&out)[NR]) {
#pragma unroll
  for (int i = 0; i < NU; i++) u[i] = add(u[i], out[i % NR]);  // keeps the loop live and the lanes reduced
}

// ---- VALU form: tab[r][i] = D_i^(r+1) in Montgomery form
__global__ __launch_bounds__(256) void valu_kernel(uint32_t* __restrict__ result, const uint32_t* __restrict__ tab, int iters) {
  uint32_t u[NU], out[NR];
  seed_lanes(u);
#pragma unroll 1
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < NR; r++) {
      uint64_t acc;
#pragma unroll
      for (int i = 0; i < NU; i++) dot_mac(acc, u[i], tab[r * NU + i], i);
      out[r] = dot_finish(acc);
    }
    if (it + 1 < iters) feed_back(u, out);
  }
#pragma unroll
  for (int r = 0; r < NR; r++) result[((size_t)blockIdx.x * 256 + threadIdx.x) * NR + r] = out[r];
}

// ---- MFMA form
constexpr int A_STRIDE = 112, O_STRIDE = 96;                       // bytes per row in LDS (96 + 16 / 80 + 16 of padding)
constexpr int WAVE_LDS = 64 * A_STRIDE + 64 * O_STRIDE;            // 13,312 bytes per wave
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void mfma_kernel(uint32_t* __restrict__ result, const v4i* __restrict__ btab, const uint32_t* __restrict__ cs, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned char* A = lds + wave * WAVE_LDS;
  unsigned char* O = A + 64 * A_STRIDE;
  uint32_t u[NU], out[NR];
  seed_lanes(u);
  int32_t c_w[7];
#pragma unroll
  for (int s = 0; s < 7; s++) c_w[s] = (int32_t)cs[s];  // below 2^31: positive as a signed factor
#pragma unroll 1
  for (int it = 0; it < iters; it++) {
    // 1. signed 8-bit limbs, four to a word: x + 0x808080 carries 128 into each of the low three bytes, the xor takes it out again
    //    as a sign: x = sum_a int8(byte_a) 2^(8a)
    uint32_t w[24];
#pragma unroll
    for (int i = 0; i < NU; i++) w[i] = (u[i] + 0x00808080u) ^ 0x00808080u;
    w[23] = 0;
#pragma unroll
    for (int q = 0; q < 6; q++) *(uint4*)(A + lane * A_STRIDE + 16 * q) = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's own writes have landed (one wave owns this LDS region)
#pragma unroll 1
    for (int t = 0; t < 4; t++) {  // 16 rows at a time
      // A operand: row 16 t + (lane & 15), limb bytes 64 c + 16 (lane >> 4) .. + 15; bytes past 96 are zero
      v4i a0 = *(const v4i*)(A + (16 * t + (lane & 15)) * A_STRIDE + 16 * (lane >> 4));
      v4i a1 = (lane >> 4) < 2 ? *(const v4i*)(A + (16 * t + (lane & 15)) * A_STRIDE + 64 + 16 * (lane >> 4)) : v4i{0, 0, 0, 0};
#pragma unroll 1
      for (int nt = 0; nt < 2; nt++) {  // 16 rounds at a time: seven accumulator tiles live, not fourteen
        v4i acc[7];
#pragma unroll
        for (int s = 0; s < 7; s++) {
          const v4i b0 = btab[((s * 2 + nt) * 2 + 0) * 64 + lane], b1 = btab[((s * 2 + nt) * 2 + 1) * 64 + lane];
          const v4i c = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b0, v4i{0, 0, 0, 0}, 0, 0, 0);
          acc[s] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, b1, c, 0, 0, 0);
        }
        // C layout: lane holds rows 16 t + 4 (lane >> 4) + j, round 16 nt + (lane & 15).  sum_s S_s 2^(8s) mod p with Montgomery-scaled
        // weights: T = offset + sum_s S_s (2^(8s) 2^32 mod p) stays in [0, 4 p^2) and reduce64(T) = T / 2^32 mod p
        const uint32_t round = 16 * nt + (lane & 15);
#pragma unroll
        for (int j = 0; j < 4; j++) {
          int64_t T = (int64_t)P << 25;
#pragma unroll
          for (int s = 0; s < 7; s++) T += (int64_t)acc[s][j] * (int64_t)c_w[s];  // v_mad_i64_i32 (plain C: the compiler then pads the MFMA -> VALU read hazard itself)
          const uint32_t x = reduce64((uint64_t)T);
          if (round < NR) *(uint32_t*)(O + (16 * t + 4 * (lane >> 4) + j) * O_STRIDE + 4 * round) = x;
        }
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
    for (int q = 0; q < 5; q++) {
      const uint4 v = *(const uint4*)(O + lane * O_STRIDE + 16 * q);
      out[4 * q] = v.x; out[4 * q + 1] = v.y; out[4 * q + 2] = v.z; out[4 * q + 3] = v.w;
    }
    if (it + 1 < iters) feed_back(u, out);
  }
#pragma unroll
  for (int r = 0; r < NR; r++) result[((size_t)blockIdx.x * 256 + threadIdx.x) * NR + r] = out[r];
}

static double time_ms(void (*launch)(int blocks, int iters), int blocks, int iters) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  launch(blocks, 4);
  (void)hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 3; rep++) {
    (void)hipEventRecord(a);
    launch(blocks, iters);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  return best;
}

static uint32_t *d_res_valu, *d_res_mfma, *d_tab, *d_cs;
static v4i* d_btab;
static void launch_valu(int blocks, int iters) { valu_kernel<<<blocks, 256>>>(d_res_valu, d_tab, iters); }
static void launch_mfma(int blocks, int iters) { mfma_kernel<<<blocks, 256, 4 * WAVE_LDS>>>(d_res_mfma, d_btab, d_cs, iters); }

int main() {
  P2Consts hk;
  p2_default_host(hk);  // from libr0hip.so
  // D_i^r for r = 1..20, i = 1..23: the first 23 words of each part_sigma row (Montgomery form)
  std::vector<uint32_t> tab(NR * NU), canon(NR * NU);
  const uint32_t* row = hk.part_sigma;
  for (int r = 0; r < NR; r++) {
    for (int i = 0; i < NU; i++) { tab[r * NU + i] = row[i]; canon[r * NU + i] = dec(row[i]); }
    row += P2_CELLS - 1 + (r + 1);
  }
  // B operands: byte j of lane l in tile (s, nt, c) = limb (s - a) of D_i^(n+1) with k = 64 c + 16 (l >> 4) + j = 4 i + a, n = 16 nt + (l & 15)
  std::vector<int8_t> btab((size_t)7 * 2 * 2 * 64 * 16, 0);
  for (int s = 0; s < 7; s++)
    for (int nt = 0; nt < 2; nt++)
      for (int c = 0; c < 2; c++)
        for (int l = 0; l < 64; l++)
          for (int j = 0; j < 16; j++) {
            const int k = 64 * c + 16 * (l >> 4) + j, i = k / 4, a = k % 4, n = 16 * nt + (l & 15), b = s - a;
            if (i >= NU || n >= NR || b < 0 || b > 3) continue;
            const uint32_t limbs = (canon[n * NU + i] + 0x00808080u) ^ 0x00808080u;
            btab[((((size_t)(s * 2 + nt) * 2 + c) * 64 + l) * 16) + j] = (int8_t)(limbs >> (8 * b));
          }
  uint32_t cs[7];
  for (int s = 0; s < 7; s++) cs[s] = fpow(enc(2), 8 * s);  // (2^(8s) mod p) 2^32 mod p: the Montgomery form of 2^(8s)
  const int max_blocks = 256 * 3;
  (void)hipMalloc(&d_res_valu, (size_t)max_blocks * 256 * NR * 4);
  (void)hipMalloc(&d_res_mfma, (size_t)max_blocks * 256 * NR * 4);
  (void)hipMalloc(&d_tab, tab.size() * 4);
  (void)hipMalloc(&d_btab, btab.size());
  (void)hipMalloc(&d_cs, sizeof cs);
  (void)hipMemcpy(d_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(d_btab, btab.data(), btab.size(), hipMemcpyHostToDevice);
  (void)hipMemcpy(d_cs, cs, sizeof cs, hipMemcpyHostToDevice);
  (void)hipFuncSetAttribute((const void*)mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * WAVE_LDS);
  // the two forms agree word for word (one and three iterations: the feedback path as well)
  for (int iters : {1, 3}) {
    launch_valu(256, iters);
    launch_mfma(256, iters);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(e)); return 2; }
    std::vector<uint32_t> a((size_t)256 * 256 * NR), b(a.size());
    (void)hipMemcpy(a.data(), d_res_valu, a.size() * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(b.data(), d_res_mfma, b.size() * 4, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (size_t k = 0; k < a.size(); k++) bad += a[k] != b[k];
    printf("iters %d: %zu of %zu words differ between the VALU and the MFMA form%s\n", iters, bad, a.size(), bad ? "  <-- MISMATCH" : "");
    if (bad) {
      for (size_t k = 0, shown = 0; k < a.size() && shown < 8; k++)
        if (a[k] != b[k]) { printf("  row %zu round %zu: valu %u mfma %u\n", k / NR, k % NR, a[k], b[k]); shown++; }
      size_t by_lane[64] = {0}, by_round[NR] = {0}, by_wave[4] = {0};
      for (size_t k = 0; k < a.size(); k++)
        if (a[k] != b[k]) { by_lane[(k / NR) & 63]++; by_round[k % NR]++; by_wave[((k / NR) >> 6) & 3]++; }
      printf("  mismatches by lane:");
      for (int l = 0; l < 64; l++) printf(" %zu", by_lane[l]);
      printf("\n  by round:");
      for (int r = 0; r < NR; r++) printf(" %zu", by_round[r]);
      printf("\n  by wave of the block: %zu %zu %zu %zu\n", by_wave[0], by_wave[1], by_wave[2], by_wave[3]);
      return 1;
    }
  }
  for (int w = 1; w <= 3; w++) {
    const int blocks = 256 * w, iters = 400;  // 256 threads = one wave on each SIMD of a CU; 256 CUs
    const double mv = time_ms(launch_valu, blocks, iters), mm = time_ms(launch_mfma, blocks, iters);
    printf("waves/SIMD %d   VALU dot products %8.3f ms = %8.1f SIMD-cycles per wave-iteration   MFMA form %8.3f ms = %8.1f   ratio %.2f\n", w, mv,
           mv * 1e-3 * 2.37e9 / ((double)iters * w), mm, mm * 1e-3 * 2.37e9 / ((double)iters * w), mm / mv);
  }
  return 0;
}
