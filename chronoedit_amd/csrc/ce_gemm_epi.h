// Row-contiguous second half of the 256-tile GEMM epilogues (bf16 and fp8 kernels, split-K reduce): see epi_chunks.
#pragma once
#include "ce_common.h"

#ifndef EPI_BIAS
#define EPI_BIAS 0
#define EPI_BIAS_GELU 1
#define EPI_GATE_RES 2
#define EPI_BIAS_GELU_ERF 3
#endif
#ifndef EPI_BIAS_ROW
#define EPI_BIAS_ROW 6  // C = bf16(acc + bias[m]): the bias runs along the ROWS of C (operand roles swapped: C = W.X^T)
#endif
#ifndef EPI_BIAS_T
#define EPI_BIAS_T 7    // C^T is stored: the caller's matrix is [N][ldc], element (n, m) = bf16(acc[m][n] + bias[n]) - the same V^T = W_v.X^T
#endif                  // WITHOUT swapping the operand roles: M stays the token count, whose tile count falls better on the chip (ce_gemm384.hip)

namespace {

// The second half of an epilogue pass for NCH 16-byte chunks per thread (8 bf16 of one row each, already "acc + bias" rounded
// to bf16, staged row-contiguous in LDS): activation or gated residual, then the global store.  The residual and gate loads
// of ALL the thread's chunks are issued first, unconditionally, on clamped addresses, and only the stores are predicated:
// written chunk by chunk under `if (m < M && n < N)` hipcc gave every chunk its own exec-mask branch with a load and an
// `s_waitcnt vmcnt(0)` inside - sixteen serial memory round trips per thread and tile in the gated-residual GEMMs.
template <int EPI, int NCH, bool HAS_GATE, typename RowCol>
__device__ __forceinline__ void epi_chunks_impl(const unsigned char* smem, int row_bytes, RowCol rowcol, int m0, int n0,
                                                bf16* __restrict__ C, const float* __restrict__ gate, const bf16* __restrict__ res,
                                                int M, int N, int ldc, int ldres, int gate_rows) {
  u32x4 v[NCH], rv[NCH];
  f32x4 g0[NCH], g1[NCH];
  int ms[NCH], ns[NCH];
#pragma unroll
  for (int t = 0; t < NCH; ++t) {
    int rl, cc, mr;
    rowcol(t, rl, cc, mr);  // staging row, 16-byte chunk, row inside the tile (== rl when a pass stages consecutive tile rows)
    ms[t] = m0 + mr;
    ns[t] = n0 + cc * 8;
    v[t] = *reinterpret_cast<const u32x4*>(smem + rl * row_bytes + cc * 16);
    if (EPI == EPI_GATE_RES) {
      const int mc = min(ms[t], M - 1), nc = min(ns[t], N - 8);
      rv[t] = *reinterpret_cast<const u32x4*>(res + (size_t)mc * ldres + nc);
      if (HAS_GATE) {
        const float* gp = gate + (gate_rows > 0 ? (size_t)(mc / gate_rows) * N : 0) + nc;  // per-sample gate rows
        g0[t] = *reinterpret_cast<const f32x4*>(gp);
        g1[t] = *reinterpret_cast<const f32x4*>(gp + 4);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < NCH; ++t) {
    u32x4 o = v[t];
    if (EPI == EPI_BIAS_GELU) {
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q] = pack_bf16(gelu_tanh(bf16lo(o[q])), gelu_tanh(bf16hi(o[q])));
    } else if (EPI == EPI_BIAS_GELU_ERF) {
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q] = pack_bf16(gelu_erf(bf16lo(o[q])), gelu_erf(bf16hi(o[q])));
    } else if (EPI == EPI_GATE_RES) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float ga = !HAS_GATE ? 1.0f : q < 2 ? g0[t][2 * q] : g1[t][2 * q - 4];
        const float gb = !HAS_GATE ? 1.0f : q < 2 ? g0[t][2 * q + 1] : g1[t][2 * q - 3];
        // x.float() + y * gate with both fp32 roundings of the reference (transformer_chronoedit.py:281,293): no fma contraction
        o[q] = pack_bf16(mul_then_add(bf16lo(o[q]), ga, bf16lo(rv[t][q])), mul_then_add(bf16hi(o[q]), gb, bf16hi(rv[t][q])));
      }
    }
    if (ms[t] < M && ns[t] < N) *reinterpret_cast<u32x4*>(C + (size_t)ms[t] * ldc + ns[t]) = o;
  }
}
// (the gate pointer is tested once, around everything: as a per-chunk `gate != nullptr ? loaded : 1` it became a select right
// behind every load, i.e. a wait per chunk again)
template <int EPI, int NCH, typename RowCol>
__device__ __forceinline__ void epi_chunks(const unsigned char* smem, int row_bytes, RowCol rowcol, int m0, int n0,
                                           bf16* __restrict__ C, const float* __restrict__ gate, const bf16* __restrict__ res,
                                           int M, int N, int ldc, int ldres, int gate_rows) {
  if (EPI == EPI_GATE_RES && gate != nullptr)
    epi_chunks_impl<EPI, NCH, true>(smem, row_bytes, rowcol, m0, n0, C, gate, res, M, N, ldc, ldres, gate_rows);
  else
    epi_chunks_impl<EPI, NCH, false>(smem, row_bytes, rowcol, m0, n0, C, gate, res, M, N, ldc, ldres, gate_rows);
}

}  // namespace
