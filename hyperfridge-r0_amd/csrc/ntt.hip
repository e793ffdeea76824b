// Batched BabyBear NTTs for gfx950: LDS-staged radix-2 butterflies, two HBM passes for sizes above 2^12.
// Replaces risc0-zkp 3.0.4 hal `batch_interpolate_ntt` / `batch_expand_into_evaluate_ntt` / `batch_bit_reverse` /
// `zk_shift` (core/ntt.rs; CUDA side in risc0-sys 1.5.0) -- SURVEY.md 8(a) a2-a5.
//
// Decomposition (N = 2^n = N1 * N2, N2 = 2^L contiguous, N1 = 2^H strided):
//   forward  (bit-reversed coeffs -> natural evals, DIT):  local size-2^L DITs on contiguous chunks, then for every
//            residue lo a size-2^H DIT over the chunk index with the inter-pass twiddle w_N^(brev_H(hi) * lo) on load;
//   inverse  (natural evals -> bit-reversed coeffs, DIF):  the transpose: strided DIF first (twiddle on store), then
//            local DIFs with the 1/N normalisation folded into the final store.
// The strided pass stages a [2^H][T] tile (T consecutive residues, 64-byte rows at T=16) so every global access is a
// run of T words; the local pass moves whole contiguous chunks.  Column-major batches map to blockIdx.y.
#include "internal.hpp"

namespace r0h {

struct TwTables {
  const uint32_t* lo;    // w22^i
  const uint32_t* hi;    // w22^(i << 11)
  const uint32_t* tw12;  // ROU[12]^i
};

__device__ __forceinline__ uint32_t omega_n(const TwTables& t, uint32_t e, uint32_t n) {
  uint32_t E = e << (MAX_DOMAIN_PO2 - n);
  return mul(t.lo[E & (TW_SIZE - 1)], t.hi[E >> TW_BITS]);
}

// One contiguous chunk of 2^L words per block.  DIR 0: DIT layers (expand_bits, L]; DIR 1: DIF layers L..1.
template <int DIR>
__global__ __launch_bounds__(256) void ntt_local_kernel(uint32_t* __restrict__ out, const uint32_t* __restrict__ in,
                                                         uint32_t L, uint32_t n_out, uint32_t expand_bits,
                                                         const uint32_t* __restrict__ tw12, uint32_t scale) {
  extern __shared__ uint32_t s[];
  const uint32_t size = 1u << L, tid = threadIdx.x;
  const size_t col = blockIdx.y;
  const uint32_t* src = in + (col << (n_out - expand_bits));
  uint32_t* dst = out + (col << n_out);
  const uint32_t base = blockIdx.x << L;
  for (uint32_t i = tid; i < size; i += 256) s[i] = src[(base + i) >> expand_bits];
  __syncthreads();
  if (DIR == 0) {
    for (uint32_t l = expand_bits + 1; l <= L; l++) {
      const uint32_t half = 1u << (l - 1);
      for (uint32_t b = tid; b < size / 2; b += 256) {
        uint32_t j = b & (half - 1), i0 = ((b >> (l - 1)) << l) + j;
        uint32_t a = s[i0], t = mul(s[i0 + half], tw12[j << (12 - l)]);
        s[i0] = add(a, t);
        s[i0 + half] = sub(a, t);
      }
      __syncthreads();
    }
  } else {
    for (uint32_t l = L; l >= 1; l--) {
      const uint32_t half = 1u << (l - 1);
      for (uint32_t b = tid; b < size / 2; b += 256) {
        uint32_t j = b & (half - 1), i0 = ((b >> (l - 1)) << l) + j;
        uint32_t a = s[i0], t = s[i0 + half];
        s[i0] = add(a, t);
        s[i0 + half] = mul(sub(a, t), tw12[j << (12 - l)]);
      }
      __syncthreads();
    }
  }
  if (DIR == 1) {
    for (uint32_t i = tid; i < size; i += 256) dst[base + i] = mul(s[i], scale);
  } else {
    for (uint32_t i = tid; i < size; i += 256) dst[base + i] = s[i];
  }
}

// A [2^H][T] tile per block: T = 2^tlog consecutive residues lo, every chunk index hi.
template <int DIR>
__global__ __launch_bounds__(512) void ntt_strided_kernel(uint32_t* __restrict__ io, uint32_t n, uint32_t L, uint32_t H,
                                                           uint32_t tlog, TwTables tw) {
  extern __shared__ uint32_t s[];
  const uint32_t T = 1u << tlog, tid = threadIdx.x, total = T << H;
  uint32_t* col = io + ((size_t)blockIdx.y << n);
  const uint32_t lo0 = blockIdx.x << tlog;
  for (uint32_t idx = tid; idx < total; idx += 512) {
    uint32_t hi = idx >> tlog, lo = lo0 + (idx & (T - 1));
    uint32_t v = col[((size_t)hi << L) + lo];
    if (DIR == 0) v = mul(v, omega_n(tw, bitrev(hi, H) * lo, n));
    s[idx] = v;
  }
  __syncthreads();
  if (DIR == 0) {
    for (uint32_t l = 1; l <= H; l++) {
      const uint32_t half = 1u << (l - 1);
      for (uint32_t b = tid; b < total / 2; b += 512) {
        uint32_t lo = b & (T - 1), jj = b >> tlog, j = jj & (half - 1);
        uint32_t i0 = ((((jj >> (l - 1)) << l) + j) << tlog) + lo, i1 = i0 + (half << tlog);
        uint32_t a = s[i0], t = mul(s[i1], tw.tw12[j << (12 - l)]);
        s[i0] = add(a, t);
        s[i1] = sub(a, t);
      }
      __syncthreads();
    }
  } else {
    for (uint32_t l = H; l >= 1; l--) {
      const uint32_t half = 1u << (l - 1);
      for (uint32_t b = tid; b < total / 2; b += 512) {
        uint32_t lo = b & (T - 1), jj = b >> tlog, j = jj & (half - 1);
        uint32_t i0 = ((((jj >> (l - 1)) << l) + j) << tlog) + lo, i1 = i0 + (half << tlog);
        uint32_t a = s[i0], t = s[i1];
        s[i0] = add(a, t);
        s[i1] = mul(sub(a, t), tw.tw12[j << (12 - l)]);
      }
      __syncthreads();
    }
  }
  for (uint32_t idx = tid; idx < total; idx += 512) {
    uint32_t hi = idx >> tlog, lo = lo0 + (idx & (T - 1));
    uint32_t v = s[idx];
    if (DIR == 1) v = mul(v, omega_n(tw, bitrev(hi, H) * lo, n));
    col[((size_t)hi << L) + lo] = v;
  }
}

__global__ void bit_reverse_kernel(uint32_t* io, uint32_t po2) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t* col = io + ((size_t)blockIdx.y << po2);
  uint32_t j = bitrev(i, po2);
  if (i < j) {
    uint32_t a = col[i], b = col[j];
    col[i] = b;
    col[j] = a;
  }
}

__global__ void zk_shift_kernel(uint32_t* io, uint32_t po2, const uint32_t* pow3_lo, const uint32_t* pow3_hi) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t* col = io + ((size_t)blockIdx.y << po2);
  uint32_t e = bitrev(i, po2);
  col[i] = mul(col[i], mul(pow3_lo[e & (TW_SIZE - 1)], pow3_hi[e >> TW_BITS]));
}

struct Split {
  uint32_t L, H, tlog;
};
static Split split_for(uint32_t n) {
  Split sp;
  if (n <= 12) { sp.L = n; sp.H = 0; sp.tlog = 0; return sp; }
  uint32_t H = n / 2;
  if (H > 10) H = 10;
  if (n - H > 12) H = n - 12;
  sp.H = H;
  sp.L = n - H;
  sp.tlog = 14 - H;
  if (sp.tlog > sp.L) sp.tlog = sp.L;
  return sp;
}

static const char* launch_check(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return make_error("%s: launch failed: %s", what, hipGetErrorString(e));
  return nullptr;
}

}  // namespace r0h

using namespace r0h;

extern "C" {

const char* r0h_batch_interpolate_ntt(r0h_ctx* ctx, r0h_buf* io, uint32_t count, uint32_t po2) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(ctx && io, "r0h_batch_interpolate_ntt: NULL argument");
  R0H_REQUIRE(po2 >= 1 && po2 <= MAX_DOMAIN_PO2, "r0h_batch_interpolate_ntt: po2 %u outside [1, %u]", po2, MAX_DOMAIN_PO2);
  R0H_REQUIRE(((size_t)count << po2) * 4 <= io->bytes, "r0h_batch_interpolate_ntt: %u columns of 2^%u exceed the buffer", count, po2);
  if (!count) return nullptr;
  const Split sp = split_for(po2);
  const uint32_t norm = inv(enc(1u << po2));
  TwTables tw{ctx->tw_lo[1], ctx->tw_hi[1], ctx->tw12[1]};
  if (sp.H) {
    KScope ks(ctx, "ntt_strided_kernel", 8.0 * count * (double)(1u << po2));
    dim3 grid(1u << (sp.L - sp.tlog), count);
    hipLaunchKernelGGL(ntt_strided_kernel<1>, grid, dim3(512), (size_t)4 << (sp.H + sp.tlog), ctx->stream, u32(io), po2, sp.L, sp.H, sp.tlog, tw);
    R0H_TRY(launch_check("ntt_strided_kernel<inv>"));
  }
  KScope ks(ctx, "ntt_local_kernel", 8.0 * count * (double)(1u << po2));
  dim3 grid(1u << (po2 - sp.L), count);
  hipLaunchKernelGGL(ntt_local_kernel<1>, grid, dim3(256), (size_t)4 << sp.L, ctx->stream, u32(io), u32(io), sp.L, po2, 0u, tw.tw12, norm);
  return launch_check("ntt_local_kernel<inv>");
  R0H_GUARD_END
}

const char* r0h_batch_expand_into_evaluate_ntt(r0h_ctx* ctx, r0h_buf* out, const r0h_buf* in, uint32_t count,
                                               uint32_t in_po2, uint32_t expand_bits) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(ctx && out && in, "r0h_batch_expand_into_evaluate_ntt: NULL argument");
  const uint32_t n = in_po2 + expand_bits;
  R0H_REQUIRE(n >= 1 && n <= MAX_DOMAIN_PO2, "r0h_batch_expand_into_evaluate_ntt: output po2 %u outside [1, %u]", n, MAX_DOMAIN_PO2);
  R0H_REQUIRE(((size_t)count << in_po2) * 4 <= in->bytes && ((size_t)count << n) * 4 <= out->bytes,
              "r0h_batch_expand_into_evaluate_ntt: %u columns exceed the buffers", count);
  R0H_REQUIRE(out->ptr != in->ptr || expand_bits == 0, "r0h_batch_expand_into_evaluate_ntt: in-place expansion is not supported");
  if (!count) return nullptr;
  const Split sp = split_for(n);
  R0H_REQUIRE(expand_bits < sp.L, "r0h_batch_expand_into_evaluate_ntt: expand_bits %u too large for size 2^%u", expand_bits, n);
  TwTables tw{ctx->tw_lo[0], ctx->tw_hi[0], ctx->tw12[0]};
  dim3 grid(1u << (n - sp.L), count);
  {
    KScope ks(ctx, "ntt_local_kernel", 4.0 * count * ((double)(1u << n) + (double)(1u << in_po2)));
    hipLaunchKernelGGL(ntt_local_kernel<0>, grid, dim3(256), (size_t)4 << sp.L, ctx->stream, u32(out), u32(in), sp.L, n, expand_bits, tw.tw12, 0u);
  }
  R0H_TRY(launch_check("ntt_local_kernel<fwd>"));
  if (sp.H) {
    KScope ks(ctx, "ntt_strided_kernel", 8.0 * count * (double)(1u << n));
    dim3 grid2(1u << (sp.L - sp.tlog), count);
    hipLaunchKernelGGL(ntt_strided_kernel<0>, grid2, dim3(512), (size_t)4 << (sp.H + sp.tlog), ctx->stream, u32(out), n, sp.L, sp.H, sp.tlog, tw);
    R0H_TRY(launch_check("ntt_strided_kernel<fwd>"));
  }
  return nullptr;
  R0H_GUARD_END
}

const char* r0h_batch_bit_reverse(r0h_ctx* ctx, r0h_buf* io, uint32_t count, uint32_t po2) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(ctx && io, "r0h_batch_bit_reverse: NULL argument");
  R0H_REQUIRE(po2 <= MAX_DOMAIN_PO2 + 2, "r0h_batch_bit_reverse: po2 %u too large", po2);
  R0H_REQUIRE(((size_t)count << po2) * 4 <= io->bytes, "r0h_batch_bit_reverse: %u columns of 2^%u exceed the buffer", count, po2);
  if (!count || po2 == 0) return nullptr;
  uint32_t threads = po2 >= 8 ? 256 : (1u << po2);
  KScope ks(ctx, "bit_reverse_kernel", 8.0 * count * (double)(1u << po2));
  dim3 grid((1u << po2) / threads, count);
  hipLaunchKernelGGL(bit_reverse_kernel, grid, dim3(threads), 0, ctx->stream, u32(io), po2);
  return launch_check("bit_reverse_kernel");
  R0H_GUARD_END
}

const char* r0h_zk_shift(r0h_ctx* ctx, r0h_buf* io, uint32_t count, uint32_t po2) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(ctx && io, "r0h_zk_shift: NULL argument");
  R0H_REQUIRE(po2 <= MAX_DOMAIN_PO2, "r0h_zk_shift: po2 %u too large", po2);
  R0H_REQUIRE(((size_t)count << po2) * 4 <= io->bytes, "r0h_zk_shift: %u columns of 2^%u exceed the buffer", count, po2);
  if (!count) return nullptr;
  uint32_t threads = po2 >= 8 ? 256 : (1u << po2);
  KScope ks(ctx, "zk_shift_kernel", 8.0 * count * (double)(1u << po2));
  dim3 grid((1u << po2) / threads, count);
  hipLaunchKernelGGL(zk_shift_kernel, grid, dim3(threads), 0, ctx->stream, u32(io), po2, ctx->pow3_lo, ctx->pow3_hi);
  return launch_check("zk_shift_kernel");
  R0H_GUARD_END
}

}  // extern "C"
