// Streaming operations of the prover for gfx950: polynomial evaluation at extension points, FRI batching mix,
// FRI fold, extension-field scans (running products, synthetic division) and the small element-wise helpers.
// Replaces risc0-zkp 3.0.4 hal `batch_evaluate_any`, `mix_poly_coeffs`, `eltwise_{add,copy}_elem`,
// `eltwise_sum_extelem`, `gather_sample`, `scatter`, `fri_fold`, `prefix_products` and core/poly.rs `poly_divide`
// (which upstream runs on the host) -- SURVEY.md 8(a) a10, a12-a15.
// All of these are HBM-bound streams: one pass over the operand, coalesced column reads, 16-byte extension accesses.
#include <algorithm>
#include <array>
#include <map>

#include "internal.hpp"

namespace r0h {

__device__ __forceinline__ Fp4 ld4(const uint32_t* p) {
  uint4 v = *(const uint4*)p;
  return Fp4{{v.x, v.y, v.z, v.w}};
}
__device__ __forceinline__ void st4(uint32_t* p, const Fp4& v) { *(uint4*)p = make_uint4(v.e[0], v.e[1], v.e[2], v.e[3]); }

static const char* launch_ok(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return make_error("%s: launch failed: %s", what, hipGetErrorString(e));
  return nullptr;
}

// ------------------------------------------------------------------ batch_evaluate_any
// sum_i c[i] x^i = sum_hi B[hi] * (sum_lo c[hi*RL + lo] * A[lo]),  A[lo] = x^lo, B[hi] = x^(hi*RL).
// A wave owns a row of RL coefficients: each lane keeps its 16 A-values in registers, multiplies base-field
// coefficients into them (4 products per coefficient), the wave reduces, lane 0 folds in B[hi].
constexpr uint32_t EVAL_BLOCKS = 64;  // partial sums per evaluation

// A[lo] and B[hi] with A[lo] * B[hi] = x^e(hi*rl + lo): e = identity for natural-order coefficients, e = brev_n for
// bit-reversed ones (brev_n(hi*rl + lo) = brev(lo) * rows + brev(hi))
__global__ void eval_tables_kernel(uint32_t* __restrict__ tabs, const uint32_t* __restrict__ points, uint32_t rl_log, uint32_t rows_log, uint32_t bitrev_coeffs) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, rl = 1u << rl_log, rows = 1u << rows_log;
  uint32_t* tab = tabs + 4 * (size_t)blockIdx.y * (rl + rows);  // one table pair per point
  const Fp4 x = ld4(points + 4 * (size_t)blockIdx.y);
  if (i < rl) st4(tab + 4 * (size_t)i, fp4_pow(x, bitrev_coeffs ? (uint64_t)bitrev(i, rl_log) << rows_log : (uint64_t)i));
  else if (i < rl + rows) st4(tab + 4 * (size_t)i, fp4_pow(x, bitrev_coeffs ? (uint64_t)bitrev(i - rl, rows_log) : (uint64_t)(i - rl) << rl_log));
}

// A column is read ONCE and evaluated at all NP points asked of it (a register's taps sit at 1-3 points z w^-back).
// sum_i c[i] x^i = sum_lo A[lo] * (sum_hi c[hi*RL + lo] * B[hi]): a lane owns KP fixed positions lo of every row and keeps, per
// point, one 64-bit running sum per (position, extension component).  A row costs one v_mad_u64_u32 per coefficient and component
// against the wave-uniform B[hi] (scalar loads) and a correction of the high words every second row -- no cross-lane step and no
// extension product inside the loop; the A[lo] products and the wave reduction happen once per wave at the end.  (Round 1, and the
// first form of this kernel, multiplied by A[lo] first and reduced every row across the wave: 1,150 instead of ~400 SIMD cycles per
// row and point.)
template <int NP>
__global__ __launch_bounds__(256) void eval_rows_kernel(uint32_t* __restrict__ partial, const uint32_t* __restrict__ coeffs,
                                                         const uint32_t* __restrict__ which, const uint32_t* __restrict__ tabs,
                                                         uint32_t po2, uint32_t rl_log) {
  constexpr int KP = 8;  // positions per lane: rows of up to 512 coefficients
  const uint32_t rl = 1u << rl_log, rows = 1u << (po2 - rl_log);
  const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t* poly = coeffs + ((size_t)which[blockIdx.y] << po2);
  const size_t tab_stride = 4 * (size_t)(rl + rows);
  uint64_t acc[NP][KP][4];
#pragma unroll
  for (int p = 0; p < NP; p++)
#pragma unroll
    for (int k = 0; k < KP; k++)
#pragma unroll
      for (int q = 0; q < 4; q++) acc[p][k][q] = 0;
  uint32_t n_rows = 0;
#pragma unroll 1
  for (uint32_t hi = blockIdx.x * 4 + wave; hi < rows; hi += gridDim.x * 4) {
    const uint32_t* row = poly + ((size_t)hi << rl_log);
    uint32_t cf[KP];
#pragma unroll
    for (int k = 0; k < KP; k++) {
      uint32_t lo = lane + 64 * k;
      cf[k] = lo < rl ? row[lo] : 0u;
    }
    // four terms fit a 64-bit sum as they are; from then on the high words are brought below p before every second term
    const bool fix = n_rows >= 4 && (n_rows & 1) == 0;
#pragma unroll
    for (int p = 0; p < NP; p++) {
      const uint32_t* bp = tabs + p * tab_stride + 4 * (size_t)(rl + hi);  // wave-uniform address
      const uint32_t b0 = bp[0], b1 = bp[1], b2 = bp[2], b3 = bp[3];
#pragma unroll
      for (int k = 0; k < KP; k++) {
        if (fix) {
#pragma unroll
          for (int q = 0; q < 4; q++) acc[p][k][q] = ((uint64_t)reduce1((uint32_t)(acc[p][k][q] >> 32)) << 32) | (uint32_t)acc[p][k][q];
        }
        acc[p][k][0] += (uint64_t)cf[k] * b0; acc[p][k][1] += (uint64_t)cf[k] * b1;
        acc[p][k][2] += (uint64_t)cf[k] * b2; acc[p][k][3] += (uint64_t)cf[k] * b3;
      }
    }
    n_rows++;
  }
  __shared__ uint32_t red[4][NP][4];
#pragma unroll
  for (int p = 0; p < NP; p++) {
    Fp4 tot = fp4_zero();
#pragma unroll
    for (int k = 0; k < KP; k++) {
      const uint32_t lo = lane + 64 * k;
      Fp4 v;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        uint64_t t = ((uint64_t)reduce1((uint32_t)(acc[p][k][q] >> 32)) << 32) | (uint32_t)acc[p][k][q];  // high word below 2p by the invariant
        v.e[q] = reduce64(t);
      }
      if (lo < rl) tot = tot + v * ld4(tabs + p * tab_stride + 4 * (size_t)lo);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      Fp4 o;
#pragma unroll
      for (int q = 0; q < 4; q++) o.e[q] = __shfl_xor(tot.e[q], off);
      tot = tot + o;
    }
    if (lane == 0) {
#pragma unroll
      for (int q = 0; q < 4; q++) red[wave][p][q] = tot.e[q];
    }
  }
  __syncthreads();
  if (threadIdx.x < NP) {
    const uint32_t p = threadIdx.x;
    Fp4 r = fp4_zero();
    for (int w = 0; w < 4; w++) r = r + Fp4{{red[w][p][0], red[w][p][1], red[w][p][2], red[w][p][3]}};
    st4(partial + 4 * (((size_t)blockIdx.y * NP + p) * gridDim.x + blockIdx.x), r);
  }
}

__global__ void eval_reduce_kernel(uint32_t* __restrict__ out, const uint32_t* __restrict__ partial,
                                   const uint32_t* __restrict__ dest, uint32_t n_partial, uint32_t n) {
  uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  Fp4 r = fp4_zero();
  for (uint32_t b = 0; b < n_partial; b++) r = r + ld4(partial + 4 * ((size_t)k * n_partial + b));
  st4(out + 4 * (size_t)dest[k], r);
}

// ------------------------------------------------------------------ mix_poly_coeffs
// combos[combo][i] += sum_{c in combo} mixpow[c] * input[c][i]; columns arrive sorted by combo.
__global__ __launch_bounds__(256) void mix_poly_kernel(uint32_t* __restrict__ combos, const uint32_t* __restrict__ input,
                                                        const uint32_t* __restrict__ params, uint32_t n_groups, uint32_t po2) {
  // params: [n_groups+1 group starts][per sorted column: col index][per group: combo id][per sorted column: 4 words mixpow]
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  const uint32_t* start = params;
  const uint32_t total = start[n_groups];
  const uint32_t* cols = params + n_groups + 1;
  const uint32_t* combo_id = cols + total;
  const uint32_t* pw = combo_id + n_groups;
  for (uint32_t g = 0; g < n_groups; g++) {
    Fp4 acc = fp4_zero();
    for (uint32_t k = start[g]; k < start[g + 1]; k++) {
      Fp4 m = Fp4{{pw[4 * k], pw[4 * k + 1], pw[4 * k + 2], pw[4 * k + 3]}};
      acc = acc + scale(m, input[((size_t)cols[k] << po2) + i]);
    }
    uint32_t* dst = combos + 4 * (((size_t)combo_id[g] << po2) + i);
    st4(dst, ld4(dst) + acc);
  }
}

// ------------------------------------------------------------------ small element-wise kernels
__global__ void add_elem_kernel(uint32_t* out, const uint32_t* a, const uint32_t* b, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = add(a[i], b[i]);
}
__global__ void copy_elem_kernel(uint32_t* out, const uint32_t* in, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i];
}
// risc0 marks untouched witness cells with Elem::INVALID (0xffffffff); the prover zeroes them before committing
__global__ void zeroize_elem_kernel(uint32_t* io, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && io[i] == 0xffffffffu) io[i] = 0u;
}
__global__ void sum_extelem_kernel(uint32_t* __restrict__ out, const uint32_t* __restrict__ in, uint32_t count, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fp4 tot = fp4_zero();
  for (uint32_t c = 0; c < count; c++) tot = tot + ld4(in + 4 * ((size_t)c * n + i));
#pragma unroll
  for (int k = 0; k < 4; k++) out[(size_t)k * n + i] = tot.e[k];
}
__global__ void gather_sample_kernel(uint32_t* dst, const uint32_t* src, uint32_t idx, uint32_t size, uint32_t stride) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < size) dst[i] = src[(size_t)i * stride + idx];
}
__global__ void scatter_kernel(uint32_t* into, const uint32_t* offsets, const uint32_t* values, uint32_t begin, uint32_t end) {
  uint32_t k = begin + blockIdx.x * blockDim.x + threadIdx.x;
  if (k < end) into[offsets[k]] = values[k];
}

// ------------------------------------------------------------------ fri_fold
__global__ void fri_fold_kernel(uint32_t* __restrict__ out, const uint32_t* __restrict__ in, Fp4 mix, uint32_t n_out) {
  uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_out) return;
  const size_t n_in = (size_t)n_out * R0H_FRI_FOLD;
  Fp4 tot = fp4_zero(), cur = fp4_one();
#pragma unroll
  for (uint32_t j = 0; j < R0H_FRI_FOLD; j++) {
    size_t src = (size_t)(__brev(j) >> 28) * n_out + idx;
    Fp4 v = Fp4{{in[src], in[n_in + src], in[2 * n_in + src], in[3 * n_in + src]}};
    tot = tot + cur * v;
    cur = cur * mix;
  }
#pragma unroll
  for (int k = 0; k < 4; k++) out[(size_t)k * n_out + idx] = tot.e[k];
}

// ------------------------------------------------------------------ extension-field scans
// Both scans use chunks of `threads * E` elements: per-chunk summary, a serial pass over the (few) chunk summaries,
// then the in-chunk pass with a Hillis-Steele scan across the block's threads.
struct ScanGeom {
  uint32_t threads, E, chunk, n_chunks;
};
static ScanGeom scan_geom(uint32_t n) {
  ScanGeom g;
  g.threads = n < 256 ? n : 256;
  g.chunk = n < 4096 ? n : 4096;
  g.E = g.chunk / g.threads;
  g.n_chunks = n / g.chunk;
  return g;
}

// running product (SUM = false): io[i] = prod_{j<=i} io[j]; running sum (SUM = true): io[i] = sum_{j<=i} io[j]
template <bool SUM> R0H_HD Fp4 scan_id() { return SUM ? fp4_zero() : fp4_one(); }
template <bool SUM> R0H_HD Fp4 scan_op(const Fp4& a, const Fp4& b) { return SUM ? a + b : a * b; }
template <bool SUM>
__global__ void prefix_chunk_prod_kernel(uint32_t* __restrict__ chunk_prod, const uint32_t* __restrict__ io, uint32_t E) {
  extern __shared__ uint32_t sh[];
  const uint32_t t = threadIdx.x, nt = blockDim.x;
  const uint32_t* p = io + 4 * ((size_t)blockIdx.x * nt * E + (size_t)t * E);
  Fp4 v = scan_id<SUM>();
  for (uint32_t k = 0; k < E; k++) v = scan_op<SUM>(v, ld4(p + 4 * k));
  st4(sh + 4 * t, v);
  __syncthreads();
  for (uint32_t s = nt / 2; s >= 1; s >>= 1) {
    if (t < s) st4(sh + 4 * t, scan_op<SUM>(ld4(sh + 4 * t), ld4(sh + 4 * (t + s))));
    __syncthreads();
  }
  if (t == 0) st4(chunk_prod + 4 * (size_t)blockIdx.x, ld4(sh));
}
// exclusive scan over the chunk summaries, one block: serial per thread, Hillis-Steele across threads
template <bool SUM>
__global__ __launch_bounds__(256) void prefix_chunk_scan_kernel(uint32_t* chunk_prod, uint32_t n_chunks) {
  __shared__ uint32_t sh[256 * 4];
  const uint32_t t = threadIdx.x, per = (n_chunks + 255) / 256, b0 = t * per;
  Fp4 v = scan_id<SUM>();
  for (uint32_t k = 0; k < per && b0 + k < n_chunks; k++) v = scan_op<SUM>(v, ld4(chunk_prod + 4 * (size_t)(b0 + k)));
  st4(sh + 4 * t, v);
  __syncthreads();
  for (uint32_t s = 1; s < 256; s <<= 1) {
    Fp4 o = t >= s ? ld4(sh + 4 * (t - s)) : scan_id<SUM>();
    __syncthreads();
    if (t >= s) st4(sh + 4 * t, scan_op<SUM>(ld4(sh + 4 * t), o));
    __syncthreads();
  }
  Fp4 cur = t ? ld4(sh + 4 * (t - 1)) : scan_id<SUM>();
  for (uint32_t k = 0; k < per && b0 + k < n_chunks; k++) {
    Fp4 x = ld4(chunk_prod + 4 * (size_t)(b0 + k));
    st4(chunk_prod + 4 * (size_t)(b0 + k), cur);
    cur = scan_op<SUM>(cur, x);
  }
}
template <bool SUM>
__global__ void prefix_apply_kernel(uint32_t* __restrict__ io, const uint32_t* __restrict__ chunk_excl, uint32_t E) {
  extern __shared__ uint32_t sh[];
  const uint32_t t = threadIdx.x, nt = blockDim.x;
  uint32_t* p = io + 4 * ((size_t)blockIdx.x * nt * E + (size_t)t * E);
  Fp4 v = scan_id<SUM>();
  for (uint32_t k = 0; k < E; k++) v = scan_op<SUM>(v, ld4(p + 4 * k));
  st4(sh + 4 * t, v);
  __syncthreads();
  for (uint32_t s = 1; s < nt; s <<= 1) {  // inclusive scan over threads
    Fp4 o = t >= s ? ld4(sh + 4 * (t - s)) : scan_id<SUM>();
    __syncthreads();
    if (t >= s) st4(sh + 4 * t, scan_op<SUM>(ld4(sh + 4 * t), o));
    __syncthreads();
  }
  Fp4 cur = ld4(chunk_excl + 4 * (size_t)blockIdx.x);
  if (t > 0) cur = scan_op<SUM>(cur, ld4(sh + 4 * (t - 1)));
  for (uint32_t k = 0; k < E; k++) {
    cur = scan_op<SUM>(cur, ld4(p + 4 * k));
    st4(p + 4 * k, cur);
  }
}

// synthetic division by (x - z): q[i] = sum_{j>i} p[j] z^(j-i-1), remainder = sum_j p[j] z^j
struct DivPowers {
  Fp4 z, zE, zChunk;  // z, z^E, z^chunk
};
// One division job = (polynomial index within a buffer of n-coefficient polynomials, point, slot of its chunk values).  The
// kernels take a batch of jobs on blockIdx.y -- the DEEP step divides ~10 combos by one to three points each: one launch set
// per "k-th point of every combo" instead of one per (combo, point), and the remainders are read back once at the end.
struct DivJob {
  DivPowers pw;
  uint32_t poly, slot, pad0, pad1;
};
__global__ void divide_chunk_sum_kernel(uint32_t* __restrict__ chunk_vals, const uint32_t* __restrict__ polys, const DivJob* __restrict__ jobs, uint32_t E,
                                        uint32_t n, uint32_t n_chunks) {
  extern __shared__ uint32_t sh[];
  const uint32_t t = threadIdx.x, nt = blockDim.x;
  const DivJob job = jobs[blockIdx.y];
  const DivPowers pw = job.pw;
  uint32_t* chunk_val = chunk_vals + 4 * (size_t)job.slot * (n_chunks + 1);
  const uint32_t* p = polys + 4 * ((size_t)job.poly * n + (size_t)blockIdx.x * nt * E + (size_t)t * E);
  Fp4 v = fp4_zero();
  for (uint32_t k = E; k-- > 0;) v = v * pw.z + ld4(p + 4 * k);  // sum_k p[k] z^k
  st4(sh + 4 * t, v);
  __syncthreads();
  // V = sum_t v_t z^(E t): pairwise combine with growing powers
  Fp4 step = pw.zE;
  for (uint32_t s = 1; s < nt; s <<= 1) {
    if ((t & (2 * s - 1)) == 0) st4(sh + 4 * t, ld4(sh + 4 * t) + ld4(sh + 4 * (t + s)) * step);
    step = step * step;
    __syncthreads();
  }
  if (t == 0) st4(chunk_val + 4 * (size_t)blockIdx.x, ld4(sh));
}
// carry[b] = sum_{b' > b} S_b' z^((b'-b-1)*chunk); remainder = sum_b S_b z^(b*chunk) -> chunk_val[n_chunks].
// One block: each thread owns `per` consecutive chunks, a suffix scan with growing powers links the threads.
__global__ __launch_bounds__(256) void divide_chunk_carry_kernel(uint32_t* chunk_vals, const DivJob* __restrict__ jobs, uint32_t n_chunks) {
  __shared__ uint32_t sh[256 * 4];
  const DivJob job = jobs[blockIdx.x];
  const DivPowers pw = job.pw;
  uint32_t* chunk_val = chunk_vals + 4 * (size_t)job.slot * (n_chunks + 1);
  const uint32_t t = threadIdx.x, per = (n_chunks + 255) / 256, b0 = t * per;
  Fp4 v = fp4_zero();  // sum_k S[b0+k] zc^k
  for (uint32_t k = per; k-- > 0;)
    if (b0 + k < n_chunks) v = v * pw.zChunk + ld4(chunk_val + 4 * (size_t)(b0 + k));
    else v = v * pw.zChunk;
  st4(sh + 4 * t, v);
  __syncthreads();
  Fp4 step = fp4_pow(pw.zChunk, per);
  for (uint32_t s = 1; s < 256; s <<= 1) {  // sh[t] = sum_{t' >= t} v_t' step^(t'-t)
    Fp4 o = t + s < 256 ? ld4(sh + 4 * (t + s)) : fp4_zero();
    __syncthreads();
    if (t + s < 256) st4(sh + 4 * t, ld4(sh + 4 * t) + o * step);
    step = step * step;
    __syncthreads();
  }
  if (t == 0) st4(chunk_val + 4 * (size_t)n_chunks, ld4(sh));  // remainder (slot past the carries)
  // carry entering this thread's top chunk from the threads above
  Fp4 cur = t + 1 < 256 ? ld4(sh + 4 * (t + 1)) : fp4_zero();
  for (uint32_t k = per; k-- > 0;) {
    if (b0 + k >= n_chunks) { cur = cur * pw.zChunk; continue; }
    Fp4 sv = ld4(chunk_val + 4 * (size_t)(b0 + k));
    st4(chunk_val + 4 * (size_t)(b0 + k), cur);
    cur = cur * pw.zChunk + sv;
  }
}
__global__ void divide_apply_kernel(uint32_t* __restrict__ polys, const uint32_t* __restrict__ chunk_vals, const DivJob* __restrict__ jobs, uint32_t E, uint32_t n,
                                    uint32_t n_chunks) {
  extern __shared__ uint32_t sh[];
  const uint32_t t = threadIdx.x, nt = blockDim.x;
  const DivJob job = jobs[blockIdx.y];
  const DivPowers pw = job.pw;
  const uint32_t* chunk_carry = chunk_vals + 4 * (size_t)job.slot * (n_chunks + 1);
  uint32_t* p = polys + 4 * ((size_t)job.poly * n + (size_t)blockIdx.x * nt * E + (size_t)t * E);
  Fp4 v = fp4_zero();
  for (uint32_t k = E; k-- > 0;) v = v * pw.z + ld4(p + 4 * k);
  st4(sh + 4 * t, v);
  __syncthreads();
  // suffix scan over threads: after it sh[t] = sum_{t' >= t} v_t' z^(E (t'-t))
  Fp4 step = pw.zE;
  for (uint32_t s = 1; s < nt; s <<= 1) {
    Fp4 o = t + s < nt ? ld4(sh + 4 * (t + s)) : fp4_zero();
    __syncthreads();
    if (t + s < nt) st4(sh + 4 * t, ld4(sh + 4 * t) + o * step);
    step = step * step;
    __syncthreads();
  }
  // carry entering this thread's range from above: threads above it, then the chunks above the block
  Fp4 zpow = fp4_pow(pw.zE, nt - 1 - t);
  Fp4 cur = ld4(chunk_carry + 4 * (size_t)blockIdx.x) * zpow;
  if (t + 1 < nt) cur = cur + ld4(sh + 4 * (t + 1));
  for (uint32_t k = E; k-- > 0;) {
    Fp4 next = cur * pw.z + ld4(p + 4 * k);
    st4(p + 4 * k, cur);
    cur = next;
  }
}

// remainders of a batch, packed: out[job] = chunk_vals[slot(job)][n_chunks]
__global__ void divide_remainders_kernel(uint32_t* __restrict__ out, const uint32_t* __restrict__ chunk_vals, uint32_t n_jobs, uint32_t n_chunks) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n_jobs) st4(out + 4 * (size_t)j, ld4(chunk_vals + 4 * ((size_t)j * (n_chunks + 1) + n_chunks)));
}

static Fp4 host_fp4(const uint32_t v[4]) { return Fp4{{v[0], v[1], v[2], v[3]}}; }
static bool canonical4(const uint32_t v[4]) { return v[0] < P && v[1] < P && v[2] < P && v[3] < P; }

}  // namespace r0h

using namespace r0h;

extern "C" {

const char* r0h_batch_evaluate_any(r0h_ctx* ctx, const r0h_buf* coeffs, uint32_t po2, const uint32_t* which,
                                   const uint32_t* xs, uint32_t n_eval, r0h_buf* out) {
  return r0h::evaluate_any(ctx, coeffs, po2, which, xs, n_eval, out, false);
}

}  // extern "C"

namespace r0h {
const char* evaluate_any(r0h_ctx* ctx, const r0h_buf* coeffs, uint32_t po2, const uint32_t* which, const uint32_t* xs,
                         uint32_t n_eval, r0h_buf* out, bool bitrev_coeffs) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(ctx && coeffs && out && (n_eval == 0 || (which && xs)), "r0h_batch_evaluate_any: NULL argument");
  R0H_REQUIRE(po2 <= MAX_DOMAIN_PO2, "r0h_batch_evaluate_any: po2 %u too large", po2);
  R0H_REQUIRE((size_t)n_eval * 16 <= out->bytes, "r0h_batch_evaluate_any: %u results exceed the output buffer", n_eval);
  if (!n_eval) return nullptr;
  const size_t n_polys = coeffs->bytes >> (po2 + 2);
  // evaluations of one column share a single read of it: group the requests by column, then the columns by the list of points
  // asked of them (in the prover: one list per tap combo), at most MAX_NP points per pass
  constexpr uint32_t MAX_NP = 2;  // 64 VGPRs of running sums per point: a third point takes a second pass over its column
  struct ColJob { uint32_t col; std::vector<uint32_t> dest; };
  std::map<uint32_t, std::vector<uint32_t>> by_col;  // column -> request indices
  for (uint32_t k = 0; k < n_eval; k++) {
    R0H_REQUIRE(which[k] < n_polys, "r0h_batch_evaluate_any: which[%u] = %u but the buffer holds %zu polynomials", k, which[k], n_polys);
    for (int q = 0; q < 4; q++) R0H_REQUIRE(xs[4 * k + q] < P, "r0h_batch_evaluate_any: xs[%u] not canonical", k);
    by_col[which[k]].push_back(k);
  }
  std::map<std::vector<uint32_t>, std::vector<ColJob>> groups;  // point list (4 words per point) -> columns
  for (auto& c : by_col)
    for (size_t at = 0; at < c.second.size(); at += MAX_NP) {
      std::vector<uint32_t> key;
      ColJob job{c.first, {}};
      for (size_t i = at; i < c.second.size() && i < at + MAX_NP; i++) {
        key.insert(key.end(), xs + 4 * (size_t)c.second[i], xs + 4 * (size_t)c.second[i] + 4);
        job.dest.push_back(c.second[i]);
      }
      groups[key].push_back(std::move(job));
    }
  const uint32_t rl_log = po2 < 9 ? po2 : 9, rl = 1u << rl_log, rows = 1u << (po2 - rl_log);
  uint32_t blocks = (rows + 3) / 4;
  if (blocks > EVAL_BLOCKS) blocks = EVAL_BLOCKS;
  size_t max_cols = 0;
  for (auto& g : groups) max_cols = std::max(max_cols, g.second.size());
  const size_t tab_words = 4 * (size_t)MAX_NP * (rl + rows), pt_words = 4 * MAX_NP, idx_words = (size_t)(1 + MAX_NP) * max_cols,
               part_words = 4 * (size_t)MAX_NP * max_cols * blocks;
  R0H_TRY(ensure_scratch(ctx, (tab_words + pt_words + idx_words + part_words) * 4));
  uint32_t* tab = (uint32_t*)ctx->scratch;
  uint32_t* pts = tab + tab_words;
  uint32_t* idx = pts + pt_words;
  uint32_t* part = idx + idx_words;
  for (auto& g : groups) {
    const uint32_t np = (uint32_t)(g.first.size() / 4), ng = (uint32_t)g.second.size();
    std::vector<uint32_t> host((size_t)(1 + np) * ng);  // [column per job][destination per (job, point)]
    for (uint32_t j = 0; j < ng; j++) {
      host[j] = g.second[j].col;
      for (uint32_t p = 0; p < np; p++) host[ng + (size_t)j * np + p] = g.second[j].dest[p];
    }
    R0H_TRY(stage_h2d(ctx, idx, host.data(), host.size() * 4));
    R0H_TRY(stage_h2d(ctx, pts, g.first.data(), g.first.size() * 4));
    KScope ks(ctx, "batch_evaluate_any", 4.0 * ng * (double)(1u << po2));  // algorithmic bytes: every column once
    // blocks per polynomial: enough workgroups overall (~2048) to fill the chip, but no more -- every block re-loads the power tables
    uint32_t gb = 2048 / ng;
    gb = gb < 4 ? 4 : gb;
    gb = gb > blocks ? blocks : gb;
    hipLaunchKernelGGL(eval_tables_kernel, dim3((rl + rows + 255) / 256, np), dim3(256), 0, ctx->stream, tab, pts, rl_log, po2 - rl_log, bitrev_coeffs ? 1u : 0u);
    if (np == 1) hipLaunchKernelGGL(eval_rows_kernel<1>, dim3(gb, ng), dim3(256), 0, ctx->stream, part, u32(coeffs), idx, tab, po2, rl_log);
    else hipLaunchKernelGGL(eval_rows_kernel<2>, dim3(gb, ng), dim3(256), 0, ctx->stream, part, u32(coeffs), idx, tab, po2, rl_log);
    hipLaunchKernelGGL(eval_reduce_kernel, dim3((ng * np + 63) / 64), dim3(64), 0, ctx->stream, u32(out), part, idx + ng, gb, ng * np);
    R0H_TRY(launch_ok("batch_evaluate_any kernels"));  // scratch reuse by the next group is ordered by the stream
  }
  return nullptr;
  R0H_GUARD_END
}
}  // namespace r0h

extern "C" {

const char* r0h_mix_poly_coeffs(r0h_ctx* ctx, r0h_buf* combos, const uint32_t mix_start[4], const uint32_t mix[4],
                                const r0h_buf* input, const uint32_t* combo_of, uint32_t input_count, uint32_t po2) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(ctx && combos && input && mix_start && mix && (combo_of || !input_count), "r0h_mix_poly_coeffs: NULL argument");
  R0H_REQUIRE(po2 <= MAX_DOMAIN_PO2, "r0h_mix_poly_coeffs: po2 %u too large", po2);
  R0H_REQUIRE(canonical4(mix_start) && canonical4(mix), "r0h_mix_poly_coeffs: mix words must be canonical (< p)");
  R0H_REQUIRE(((size_t)input_count << po2) * 4 <= input->bytes, "r0h_mix_poly_coeffs: %u columns exceed the input buffer", input_count);
  if (!input_count) return nullptr;
  std::vector<uint32_t> order(input_count);
  for (uint32_t c = 0; c < input_count; c++) {
    R0H_REQUIRE((((size_t)combo_of[c] + 1) << po2) * 16 <= combos->bytes, "r0h_mix_poly_coeffs: combo %u outside the combos buffer", combo_of[c]);
    order[c] = c;
  }
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return combo_of[a] < combo_of[b]; });
  std::vector<Fp4> pw(input_count);
  Fp4 cur = host_fp4(mix_start), m = host_fp4(mix);
  for (uint32_t c = 0; c < input_count; c++) { pw[c] = cur; cur = cur * m; }
  std::vector<uint32_t> starts, ids;
  for (uint32_t k = 0; k < input_count; k++)
    if (k == 0 || combo_of[order[k]] != combo_of[order[k - 1]]) { starts.push_back(k); ids.push_back(combo_of[order[k]]); }
  const uint32_t n_groups = (uint32_t)ids.size();
  starts.push_back(input_count);
  std::vector<uint32_t> params;
  params.insert(params.end(), starts.begin(), starts.end());
  params.insert(params.end(), order.begin(), order.end());
  params.insert(params.end(), ids.begin(), ids.end());
  for (uint32_t k = 0; k < input_count; k++)
    for (int q = 0; q < 4; q++) params.push_back(pw[order[k]].e[q]);
  R0H_TRY(ensure_scratch(ctx, params.size() * 4));
  R0H_TRY(stage_h2d(ctx, ctx->scratch, params.data(), params.size() * 4));
  uint32_t n = 1u << po2, threads = n < 256 ? n : 256;
  KScope ks(ctx, "mix_poly_kernel", 4.0 * input_count * (double)n + 32.0 * n_groups * (double)n);
  hipLaunchKernelGGL(mix_poly_kernel, dim3(n / threads), dim3(threads), 0, ctx->stream, u32(combos), u32(input), (const uint32_t*)ctx->scratch, n_groups, po2);
  return launch_ok("mix_poly_kernel");
  R0H_GUARD_END
}

const char* r0h_eltwise_add_elem(r0h_ctx* ctx, r0h_buf* out, const r0h_buf* a, const r0h_buf* b, uint32_t n) {
  R0H_REQUIRE(ctx && out && a && b, "r0h_eltwise_add_elem: NULL argument");
  R0H_REQUIRE((size_t)n * 4 <= out->bytes && (size_t)n * 4 <= a->bytes && (size_t)n * 4 <= b->bytes, "r0h_eltwise_add_elem: n %u exceeds a buffer", n);
  if (!n) return nullptr;
  hipLaunchKernelGGL(add_elem_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, u32(out), u32(a), u32(b), n);
  return launch_ok("add_elem_kernel");
}
const char* r0h_eltwise_copy_elem(r0h_ctx* ctx, r0h_buf* out, const r0h_buf* in, uint32_t n) {
  R0H_REQUIRE(ctx && out && in, "r0h_eltwise_copy_elem: NULL argument");
  R0H_REQUIRE((size_t)n * 4 <= out->bytes && (size_t)n * 4 <= in->bytes, "r0h_eltwise_copy_elem: n %u exceeds a buffer", n);
  if (!n) return nullptr;
  hipLaunchKernelGGL(copy_elem_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, u32(out), u32(in), n);
  return launch_ok("copy_elem_kernel");
}
const char* r0h_eltwise_zeroize_elem(r0h_ctx* ctx, r0h_buf* io, uint32_t n) {
  R0H_REQUIRE(ctx && io, "r0h_eltwise_zeroize_elem: NULL argument");
  R0H_REQUIRE((size_t)n * 4 <= io->bytes, "r0h_eltwise_zeroize_elem: n %u exceeds the buffer", n);
  if (!n) return nullptr;
  hipLaunchKernelGGL(zeroize_elem_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, u32(io), n);
  return launch_ok("zeroize_elem_kernel");
}
const char* r0h_eltwise_sum_extelem(r0h_ctx* ctx, r0h_buf* out, const r0h_buf* in, uint32_t count, uint32_t n) {
  R0H_REQUIRE(ctx && out && in, "r0h_eltwise_sum_extelem: NULL argument");
  R0H_REQUIRE((size_t)n * 16 <= out->bytes && (size_t)n * count * 16 <= in->bytes, "r0h_eltwise_sum_extelem: sizes exceed a buffer");
  if (!n) return nullptr;
  hipLaunchKernelGGL(sum_extelem_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, u32(out), u32(in), count, n);
  return launch_ok("sum_extelem_kernel");
}
const char* r0h_gather_sample(r0h_ctx* ctx, r0h_buf* dst, const r0h_buf* src, uint32_t idx, uint32_t size, uint32_t stride) {
  R0H_REQUIRE(ctx && dst && src, "r0h_gather_sample: NULL argument");
  R0H_REQUIRE((size_t)size * 4 <= dst->bytes, "r0h_gather_sample: size %u exceeds dst", size);
  R0H_REQUIRE(size == 0 || ((size_t)(size - 1) * stride + idx + 1) * 4 <= src->bytes, "r0h_gather_sample: reads past src");
  if (!size) return nullptr;
  hipLaunchKernelGGL(gather_sample_kernel, dim3((size + 255) / 256), dim3(256), 0, ctx->stream, u32(dst), u32(src), idx, size, stride);
  return launch_ok("gather_sample_kernel");
}
const char* r0h_scatter(r0h_ctx* ctx, r0h_buf* into, const r0h_buf* index, const r0h_buf* offsets, const r0h_buf* values, uint32_t n_index) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(ctx && into && index && offsets && values, "r0h_scatter: NULL argument");
  if (n_index < 2) return nullptr;
  R0H_REQUIRE((size_t)n_index * 4 <= index->bytes, "r0h_scatter: n_index exceeds the index buffer");
  std::vector<uint32_t> idx(n_index);
  R0H_TRY(r0h_buf_d2h(ctx, index, 0, idx.data(), (size_t)n_index * 4));
  uint32_t begin = idx[0], end = idx[n_index - 1];
  R0H_REQUIRE(begin <= end && (size_t)end * 4 <= offsets->bytes && (size_t)end * 4 <= values->bytes, "r0h_scatter: index range [%u, %u) outside offsets/values", begin, end);
  std::vector<uint32_t> offs(end - begin);
  if (end > begin) {
    R0H_TRY(r0h_buf_d2h(ctx, offsets, (size_t)begin * 4, offs.data(), (size_t)(end - begin) * 4));
    for (uint32_t o : offs) R0H_REQUIRE((size_t)o * 4 < into->bytes, "r0h_scatter: offset %u outside the destination", o);
    hipLaunchKernelGGL(scatter_kernel, dim3((end - begin + 255) / 256), dim3(256), 0, ctx->stream, u32(into), u32(offsets), u32(values), begin, end);
  }
  return launch_ok("scatter_kernel");
  R0H_GUARD_END
}

// ---- the same operations with the operand placement of the risc0-zkp `Hal` trait (as recalled): `which`, `xs` and
// `combos` are device Buffers there, `scatter` takes host slices.  The grouping these kernels rely on is made on the host, so the
// Buffer forms read the small index buffers back (one copy of a few KB) -- a Rust `HipHal` calls these and never copies itself.
const char* r0h_batch_evaluate_any_buf(r0h_ctx* ctx, const r0h_buf* coeffs, uint32_t po2, const r0h_buf* which, const r0h_buf* xs, uint32_t n_eval, r0h_buf* out) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(ctx && coeffs && out && which && xs, "r0h_batch_evaluate_any_buf: NULL argument");
  R0H_REQUIRE((size_t)n_eval * 4 <= which->bytes && (size_t)n_eval * 16 <= xs->bytes, "r0h_batch_evaluate_any_buf: %u evaluations exceed the which / xs buffers", n_eval);
  std::vector<uint32_t> w(n_eval), x(4 * (size_t)n_eval);
  if (n_eval) {
    R0H_TRY(r0h_buf_d2h(ctx, which, 0, w.data(), w.size() * 4));
    R0H_TRY(r0h_buf_d2h(ctx, xs, 0, x.data(), x.size() * 4));
  }
  return r0h::evaluate_any(ctx, coeffs, po2, w.data(), x.data(), n_eval, out, false);
  R0H_GUARD_END
}

const char* r0h_mix_poly_coeffs_buf(r0h_ctx* ctx, r0h_buf* combos, const uint32_t mix_start[4], const uint32_t mix[4], const r0h_buf* input, const r0h_buf* combo_of,
                                    uint32_t input_count, uint32_t po2) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(ctx && combo_of, "r0h_mix_poly_coeffs_buf: NULL argument");
  R0H_REQUIRE((size_t)input_count * 4 <= combo_of->bytes, "r0h_mix_poly_coeffs_buf: %u columns exceed the combo buffer", input_count);
  std::vector<uint32_t> c(input_count);
  if (input_count) R0H_TRY(r0h_buf_d2h(ctx, combo_of, 0, c.data(), c.size() * 4));
  return r0h_mix_poly_coeffs(ctx, combos, mix_start, mix, input, c.data(), input_count, po2);
  R0H_GUARD_END
}

// Hal::scatter(into, index: &[u32], offsets: &[u32], values: &[Elem]): values[k] goes to into[offsets[k]] for index[0] <= k < index[n_index - 1]
const char* r0h_scatter_slices(r0h_ctx* ctx, r0h_buf* into, const uint32_t* index, uint32_t n_index, const uint32_t* offsets, const uint32_t* values, uint32_t n_values) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(ctx && into && (index || !n_index) && ((offsets && values) || !n_values), "r0h_scatter_slices: NULL argument");
  if (n_index < 2) return nullptr;
  const uint32_t begin = index[0], end = index[n_index - 1];
  R0H_REQUIRE(begin <= end && end <= n_values, "r0h_scatter_slices: index range [%u, %u) outside %u offsets/values", begin, end, n_values);
  if (end == begin) return nullptr;
  for (uint32_t k = begin; k < end; k++) {
    R0H_REQUIRE((size_t)offsets[k] * 4 < into->bytes, "r0h_scatter_slices: offset %u outside the destination", offsets[k]);
    R0H_REQUIRE(values[k] < P, "r0h_scatter_slices: value %u is not a canonical field word", k);
  }
  const size_t bytes = (size_t)(end - begin) * 4;
  R0H_TRY(ensure_scratch(ctx, 2 * bytes));
  uint32_t* d_off = (uint32_t*)ctx->scratch;
  uint32_t* d_val = d_off + (end - begin);
  R0H_TRY(stage_h2d(ctx, d_off, offsets + begin, bytes));
  R0H_TRY(stage_h2d(ctx, d_val, values + begin, bytes));
  hipLaunchKernelGGL(scatter_kernel, dim3((end - begin + 255) / 256), dim3(256), 0, ctx->stream, u32(into), d_off, d_val, 0u, end - begin);
  return launch_ok("scatter_kernel");
  R0H_GUARD_END
}

const char* r0h_fri_fold(r0h_ctx* ctx, r0h_buf* out, const r0h_buf* in, const uint32_t mix[4], uint32_t n_out) {
  R0H_REQUIRE(ctx && out && in && mix, "r0h_fri_fold: NULL argument");
  R0H_REQUIRE(canonical4(mix), "r0h_fri_fold: mix words must be canonical (< p)");
  R0H_REQUIRE((size_t)n_out * 16 <= out->bytes && (size_t)n_out * 16 * R0H_FRI_FOLD <= in->bytes, "r0h_fri_fold: n_out %u exceeds a buffer", n_out);
  if (!n_out) return nullptr;
  hipLaunchKernelGGL(fri_fold_kernel, dim3((n_out + 255) / 256), dim3(256), 0, ctx->stream, u32(out), u32(in), host_fp4(mix), n_out);
  return launch_ok("fri_fold_kernel");
}

}  // extern "C"
template <bool SUM>
static const char* prefix_scan(r0h_ctx* ctx, r0h_buf* io, uint32_t n, const char* what) {
  R0H_REQUIRE(ctx && io, "%s: NULL argument", what);
  R0H_REQUIRE(n && (n & (n - 1)) == 0, "%s: n %u is not a power of two", what, n);
  R0H_REQUIRE((size_t)n * 16 <= io->bytes, "%s: n %u exceeds the buffer", what, n);
  const ScanGeom g = scan_geom(n);
  R0H_TRY(ensure_scratch(ctx, (size_t)(g.n_chunks + 1) * 16));
  uint32_t* cp = (uint32_t*)ctx->scratch;
  hipLaunchKernelGGL(prefix_chunk_prod_kernel<SUM>, dim3(g.n_chunks), dim3(g.threads), g.threads * 16, ctx->stream, cp, u32(io), g.E);
  hipLaunchKernelGGL(prefix_chunk_scan_kernel<SUM>, dim3(1), dim3(256), 0, ctx->stream, cp, g.n_chunks);
  hipLaunchKernelGGL(prefix_apply_kernel<SUM>, dim3(g.n_chunks), dim3(g.threads), g.threads * 16, ctx->stream, u32(io), cp, g.E);
  return launch_ok(what);
}
extern "C" {
const char* r0h_prefix_products(r0h_ctx* ctx, r0h_buf* io, uint32_t n) { return prefix_scan<false>(ctx, io, n, "r0h_prefix_products"); }
// the additive counterpart (the running sums of a log-derivative argument): io[i] = sum_{j <= i} io[j] over extension elements
const char* r0h_prefix_sums(r0h_ctx* ctx, r0h_buf* io, uint32_t n) { return prefix_scan<true>(ctx, io, n, "r0h_prefix_sums"); }

const char* r0h_poly_divide(r0h_ctx* ctx, r0h_buf* poly, uint32_t n, const uint32_t z[4], uint32_t remainder[4]) {
  R0H_REQUIRE(ctx && poly && z, "r0h_poly_divide: NULL argument");
  R0H_REQUIRE(canonical4(z), "r0h_poly_divide: z words must be canonical (< p)");
  R0H_REQUIRE(n && (n & (n - 1)) == 0, "r0h_poly_divide: n %u is not a power of two", n);
  R0H_REQUIRE((size_t)n * 16 <= poly->bytes, "r0h_poly_divide: n %u exceeds the buffer", n);
  const uint32_t idx = 0;
  return r0h::poly_divide_batch(ctx, poly, n, &idx, z, 1, remainder);
}

}  // extern "C"

namespace r0h {
// polys: a buffer of n-coefficient extension polynomials (AoS, natural order); job j divides polynomial poly_idx[j] in place by
// (x - points[4j..4j+4)).  Jobs on the same polynomial are applied in the order given.  remainders_host (4 words per job, may be
// NULL) is filled by ONE blocking copy after everything has been enqueued.
const char* poly_divide_batch(r0h_ctx* ctx, r0h_buf* polys, uint32_t n, const uint32_t* poly_idx, const uint32_t* points, uint32_t n_jobs, uint32_t* remainders_host) {
  R0H_GUARD_BEGIN
  if (!n_jobs) return nullptr;
  const ScanGeom g = scan_geom(n);
  const size_t n_polys = polys->bytes / ((size_t)n * 16);
  // pass k holds the k-th job of every polynomial: within a pass the jobs touch different polynomials and run side by side
  std::vector<std::vector<uint32_t>> passes;
  {
    std::map<uint32_t, uint32_t> seen;
    for (uint32_t j = 0; j < n_jobs; j++) {
      R0H_REQUIRE(poly_idx[j] < n_polys, "poly_divide: polynomial %u outside a buffer of %zu", poly_idx[j], n_polys);
      const uint32_t k = seen[poly_idx[j]]++;
      if (passes.size() <= k) passes.emplace_back();
      passes[k].push_back(j);
    }
  }
  std::vector<DivJob> jobs;  // in launch order; slot = the job's index as given (remainders come back in the caller's order)
  for (const auto& pass : passes)
    for (uint32_t j : pass) {
      DivJob d;
      d.pw.z = host_fp4(points + 4 * (size_t)j);
      d.pw.zE = fp4_pow(d.pw.z, g.E);
      d.pw.zChunk = fp4_pow(d.pw.z, g.chunk);
      d.poly = poly_idx[j]; d.slot = j; d.pad0 = d.pad1 = 0;
      jobs.push_back(d);
    }
  const size_t cv_bytes = (size_t)n_jobs * (g.n_chunks + 1) * 16, job_bytes = jobs.size() * sizeof(DivJob), rem_bytes = (size_t)n_jobs * 16;
  R0H_TRY(ensure_scratch(ctx, cv_bytes + job_bytes + rem_bytes));
  uint32_t* cv = (uint32_t*)ctx->scratch;
  DivJob* d_jobs = (DivJob*)((char*)ctx->scratch + cv_bytes);
  uint32_t* d_rem = (uint32_t*)((char*)ctx->scratch + cv_bytes + job_bytes);
  R0H_TRY(stage_h2d(ctx, d_jobs, jobs.data(), job_bytes));
  KScope ks(ctx, "poly_divide", 48.0 * n * n_jobs);
  size_t first = 0;
  for (const auto& pass : passes) {
    const uint32_t nj = (uint32_t)pass.size();
    hipLaunchKernelGGL(divide_chunk_sum_kernel, dim3(g.n_chunks, nj), dim3(g.threads), g.threads * 16, ctx->stream, cv, u32(polys), d_jobs + first, g.E, n, g.n_chunks);
    hipLaunchKernelGGL(divide_chunk_carry_kernel, dim3(nj), dim3(256), 0, ctx->stream, cv, d_jobs + first, g.n_chunks);
    hipLaunchKernelGGL(divide_apply_kernel, dim3(g.n_chunks, nj), dim3(g.threads), g.threads * 16, ctx->stream, u32(polys), cv, d_jobs + first, g.E, n, g.n_chunks);
    first += nj;
  }
  R0H_TRY(launch_ok("poly_divide kernels"));
  if (remainders_host) {
    hipLaunchKernelGGL(divide_remainders_kernel, dim3((n_jobs + 63) / 64), dim3(64), 0, ctx->stream, d_rem, cv, n_jobs, g.n_chunks);
    R0H_TRY(launch_ok("divide_remainders_kernel"));
    R0H_TRY_HIP(hipMemcpyAsync(remainders_host, d_rem, rem_bytes, hipMemcpyDeviceToHost, ctx->stream));
    R0H_TRY_HIP(hipStreamSynchronize(ctx->stream));
  }
  return nullptr;
  R0H_GUARD_END
}
}  // namespace r0h
