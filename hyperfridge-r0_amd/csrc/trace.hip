// Witness generation of the trace circuit on the device: the executor's compact preflight rows (72 bytes per cycle, 16 per boundary
// row) are uploaded and one thread per trace row expands them into the column-major Montgomery DATA group -- risc0-circuit-rv32im
// 4.0.4's `generate_witness` step (SURVEY.md 3.4 step 2, 8(a) a9, 8(e) "do witgen on device from the compact preflight trace").
// A 2^20-row segment uploads 72 MiB instead of the 576 MiB the expanded group occupies, and the expansion is an HBM-bound stream:
// 72 B read + 138 x 4 B written per row, consecutive lanes own consecutive rows, so every column store is one contiguous 256-byte
// line per wave.  The expansion itself is csrc/trace.hpp, the same code the host reference (r0h_vm_trace_witness) compiles.  The two
// multiplicity columns stay zero here: r0h_logup_multiplicities counts the lookups afterwards (csrc/logup.hip).
#include "trace.hpp"

namespace r0h {

__global__ __launch_bounds__(256) void trace_witgen_kernel(uint32_t* __restrict__ data, const r0h_preflight_row* __restrict__ rows, uint32_t n_rows,
                                                           const r0h_preflight_bound* __restrict__ bounds, uint32_t n_bounds,
                                                           const trace::Tables* __restrict__ tables, uint32_t po2, uint32_t seg_number, uint32_t closing) {
  const uint32_t r = blockIdx.x * 256u + threadIdx.x, n = 1u << po2;
  if (r >= n) return;
  uint32_t* cell = data + r;
  // every column of the row is written exactly once: the buffer needs no clearing beforehand.  `put` and `raw` only remember what
  // the expansion sets; the sweep below stores the row, zeros included.
  uint32_t vals[trace::N_COLS];
#pragma unroll
  for (uint32_t c = 0; c < trace::N_COLS; c++) vals[c] = 0;
  auto put = [&](uint32_t col, uint32_t v) { vals[col] = enc(v); };
  auto raw = [&](uint32_t col, uint32_t w) { vals[col] = w; };
  if (r < n_rows) trace::live_row(rows[r], *tables, put, raw);
  else if (r < n_rows + n_bounds) {
    const uint32_t j = r - n_rows;
    trace::bound_row(bounds[j], j ? bounds[j - 1].addr : 0xffffffffu, seg_number, closing != 0, *tables, put, raw);
  } else trace::blank_row(*tables, put, raw);
#pragma unroll
  for (uint32_t c = 0; c < trace::N_COLS; c++) cell[(size_t)c << po2] = vals[c];
}

}  // namespace r0h

using namespace r0h;

extern "C" {

const char* r0h_trace_witgen(r0h_ctx* ctx, const r0h_preflight_row* rows, size_t n_rows, const r0h_preflight_bound* bounds, size_t n_bounds,
                             uint32_t po2, const r0h_trace_segment* segment, r0h_buf* data, uint32_t globals_out[R0H_TRACE_GLOBALS]) {
  R0H_GUARD_BEGIN
  R0H_REQUIRE(ctx && (rows || !n_rows) && data && (bounds || !n_bounds) && globals_out && segment, "r0h_trace_witgen: NULL argument");
  R0H_REQUIRE(n_rows + n_bounds >= 1, "r0h_trace_witgen: a segment has at least one cycle or one boundary row");
  R0H_REQUIRE(segment->number >= 1 && segment->number <= 65536, "r0h_trace_witgen: segment number %u outside [1, 65536]", segment->number);
  R0H_REQUIRE(po2 >= R0H_TRACE_MIN_PO2 && po2 <= R0H_TRACE_MAX_PO2, "r0h_trace_witgen: po2 %u outside [%u, %u] (the lookup tables have 2^16 rows)", po2, (unsigned)R0H_TRACE_MIN_PO2,
              (unsigned)R0H_TRACE_MAX_PO2);
  R0H_REQUIRE(n_rows + n_bounds <= ((size_t)1 << po2), "r0h_trace_witgen: %zu cycles and %zu boundary rows do not fit 2^%u", n_rows, n_bounds, po2);
  for (size_t j = 0; j < n_bounds; j += (n_bounds > 4096 ? n_bounds / 4096 : 1))
    R0H_REQUIRE(bounds[j].prev_seg < segment->number && bounds[j].addr < R0H_REG_BASE + 32, "r0h_trace_witgen: boundary row %zu names segment %u (own: %u) or an address outside memory and registers", j,
                bounds[j].prev_seg, segment->number);
  const size_t n = (size_t)1 << po2;
  R0H_REQUIRE((size_t)R0H_TRACE_COLUMNS * n * 4 <= data->bytes, "r0h_trace_witgen: the DATA buffer holds fewer than %u columns of 2^%u rows", (unsigned)R0H_TRACE_COLUMNS, po2);
  // what the kernel's indexing relies on, checked on the host: cycle numbers are the row numbers, timestamps are in order, a pc is
  // one field element
  for (size_t r = 0; r < n_rows; r += (n_rows > 4096 ? n_rows / 4096 : 1)) {
    R0H_REQUIRE(rows[r].cycle == r, "r0h_trace_witgen: row %zu carries cycle %u", r, rows[r].cycle);
    R0H_REQUIRE(rows[r].pc < P && rows[r].next_pc < P, "r0h_trace_witgen: pc %#x is not below p: the trace circuit carries a pc as one field element", rows[r].pc);
  }
  KScope ks(ctx, "trace_witgen", (double)n_rows * sizeof(r0h_preflight_row) + (double)n_bounds * sizeof(r0h_preflight_bound) + (double)R0H_TRACE_COLUMNS * n * 4);
  const size_t row_bytes = n_rows * sizeof(r0h_preflight_row), bound_bytes = n_bounds * sizeof(r0h_preflight_bound), tab_off = (row_bytes + bound_bytes + 15) & ~(size_t)15;
  r0h_buf* staging = nullptr;
  R0H_TRY(buf_alloc_pooled(ctx, tab_off + sizeof(trace::Tables), &staging));
  struct Free { r0h_buf* b; ~Free() { r0h_buf_free(b); } } guard{staging};
  char* base = (char*)staging->ptr;
  // the caller's arrays are pageable: the copies are stream-ordered but return only once the source has been read
  if (n_rows) R0H_TRY_HIP(hipMemcpyAsync(base, rows, row_bytes, hipMemcpyHostToDevice, ctx->stream));
  if (n_bounds) R0H_TRY_HIP(hipMemcpyAsync(base + row_bytes, bounds, bound_bytes, hipMemcpyHostToDevice, ctx->stream));
  R0H_TRY(stage_h2d(ctx, base + tab_off, &trace::trace_tables(), sizeof(trace::Tables)));
  hipLaunchKernelGGL(trace_witgen_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, ctx->stream, u32(data), (const r0h_preflight_row*)base, (uint32_t)n_rows,
                     (const r0h_preflight_bound*)(base + row_bytes), (uint32_t)n_bounds, (const trace::Tables*)(base + tab_off), po2, segment->number, segment->closing);
  hipError_t e = hipGetLastError();
  R0H_REQUIRE(e == hipSuccess, "trace_witgen_kernel: %s", hipGetErrorString(e));
  trace::trace_globals(rows, n_rows, bounds, n_bounds, segment->number, segment->closing != 0, segment->idle_pc, globals_out);
  R0H_TRY_HIP(hipStreamSynchronize(ctx->stream));  // the staging block goes back to the pool and the caller may free its rows
  return nullptr;
  R0H_GUARD_END
}

}  // extern "C"
