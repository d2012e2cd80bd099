// HBM-bound row kernels of the ChronoEdit DiT block (one wave64 per token row, 16-B loads):
//   ce_ln_affine      K5/K11/K18  LN(fp32 stats, eps) * a[d] + b[d] -> bf16
//                                 (reference: transformer_chronoedit.py:279,284,289,460)
//   ce_rmsnorm_rope   K7+K8       RMSNorm across ALL heads' channels, then 3-D RoPE on
//                                 (even,odd) pairs (reference: transformer_chronoedit.py:62-79)
// plus the once-per-forward small kernels (timestep embed K2, modulation tables, patchify,
// unpatchify).  Algorithmic bytes per row pass: 2 * D * 2 B (read bf16 + write bf16).
#include "ce_common.h"

#define ROW_MAXC 10  // 16-B chunks per lane: D <= 64 * 8 * 10 = 5120
#ifndef CE_LN_ROWS
#define CE_LN_ROWS 2  // activation rows per wave of the full-width LN-modulate kernel (A/B in profiles/r02_row_kernels_ab.txt)
#endif

// ------------------------------------------------------------------------------------
// LN * a + b
// ------------------------------------------------------------------------------------
// FP8: the bf16-rounded result is not written; it is quantised in registers to OCP e4m3 with one scale per row (the contract of
// ce_quant_rows_fp8 applied to the row this kernel would have written) - the A operand of ce_gemm_fp8 without the extra pass.
// FULL: D == 64 * 8 * ROW_MAXC, every lane owns exactly ROW_MAXC chunks and the per-chunk guards vanish.  With the guards hipcc
// wrapped every 16-byte load in its own exec-mask branch and waited for it (`global_load_dwordx4; s_waitcnt vmcnt(0)` ten
// times per row): one load in flight per wave, 3.7 TB/s by occupancy alone.
// R rows per wave: the (a, b) rows are fp32 - 40 KB of L2 reads per activation row against 10 KB read + 10 KB written of HBM traffic,
// which made the L2, not HBM, the limit of the one-row-per-wave form (4.2 TB/s).  A wave that owns R consecutive rows of one
// sample reads each (a, b) chunk once for all of them.  The launcher picks R so that a wave's rows never straddle two samples.
// FP8 == 2: the MX form - every 32 consecutive elements of the row get their own E8M0 scale (ce_quant_rows_mxfp8's contract applied to
// the bf16 row this kernel would have written); no row-wide amax, so the quantisation runs in the same sweep as the affine.
template <int FP8, bool FULL, int R>
__global__ __launch_bounds__(256) void ln_affine_kernel(const bf16* __restrict__ x, bf16* __restrict__ y,
                                                        const float* __restrict__ a, const float* __restrict__ b,
                                                        int M, int D, int ldx, int ldy, float eps, int ab_rows, int ab_stride,
                                                        unsigned char* __restrict__ q8, float* __restrict__ qscale, int ldq) {
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
  if (row0 >= M) return;
  if (ab_rows > 0) {  // one (a, b) pair per group of ab_rows rows (samples stacked along M)
    a += (size_t)(row0 / ab_rows) * ab_stride;
    b += (size_t)(row0 / ab_rows) * ab_stride;
  }
  const int nch = D >> 3;
  u32x4 raw[R][ROW_MAXC];
  float mean[R], rstd[R];
#pragma unroll
  for (int k = 0; k < R; ++k) {  // all loads of all rows first (rows past M re-read the last row; their stores are predicated)
    const bf16* xr = x + (size_t)min(row0 + k, M - 1) * ldx;
#pragma unroll
    for (int i = 0; i < ROW_MAXC; ++i) {
      const int c = lane + 64 * i;
      if (FULL || c < nch) raw[k][i] = *reinterpret_cast<const u32x4*>(xr + c * 8);
    }
  }
#pragma unroll
  for (int k = 0; k < R; ++k) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < ROW_MAXC; ++i) {
      const int c = lane + 64 * i;
      if (FULL || c < nch) {
#pragma unroll
        for (int j = 0; j < 4; ++j) s += bf16lo(raw[k][i][j]) + bf16hi(raw[k][i][j]);
      }
    }
    mean[k] = wave_sum(s) / (float)D;
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < ROW_MAXC; ++i) {
      const int c = lane + 64 * i;
      if (FULL || c < nch) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float d0 = bf16lo(raw[k][i][j]) - mean[k], d1 = bf16hi(raw[k][i][j]) - mean[k];
          v += d0 * d0 + d1 * d1;
        }
      }
    }
    rstd[k] = 1.0f / sqrtf(wave_sum(v) / (float)D + eps);
  }
  float amax[R];
#pragma unroll
  for (int k = 0; k < R; ++k) amax[k] = 0.f;
#pragma unroll
  for (int i = 0; i < ROW_MAXC; ++i) {
    const int c = lane + 64 * i;
    if (FULL || c < nch) {
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(a + c * 8), a1 = *reinterpret_cast<const f32x4*>(a + c * 8 + 4);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(b + c * 8), b1 = *reinterpret_cast<const f32x4*>(b + c * 8 + 4);
#pragma unroll
      for (int k = 0; k < R; ++k) {
        u32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float aa0 = j < 2 ? a0[2 * j] : a1[2 * j - 4], aa1 = j < 2 ? a0[2 * j + 1] : a1[2 * j - 3];
          const float bb0 = j < 2 ? b0[2 * j] : b1[2 * j - 4], bb1 = j < 2 ? b0[2 * j + 1] : b1[2 * j - 3];
          const float n0 = (bf16lo(raw[k][i][j]) - mean[k]) * rstd[k], n1 = (bf16hi(raw[k][i][j]) - mean[k]) * rstd[k];
          o[j] = pack_bf16(n0 * aa0 + bb0, n1 * aa1 + bb1);
          if (FP8 == 1) amax[k] = fmaxf(amax[k], fmaxf(fabsf(bf16lo(o[j])), fabsf(bf16hi(o[j]))));
        }
        if (FP8 == 1) raw[k][i] = o;
        else if (FP8 == 0 && row0 + k < M) *reinterpret_cast<u32x4*>(y + (size_t)(row0 + k) * ldy + c * 8) = o;
        if (FP8 == 2) raw[k][i] = o;  // (quantised below, outside the chunk guard: the block amax is a cross-lane exchange)
      }
    }
    if (FP8 == 2) {
#pragma unroll
      for (int k = 0; k < R; ++k) {
        const bool on = (FULL || c < nch) && row0 + k < M;
        const u32x4 v = raw[k][i];
        float am = 0.f;
        if (FULL || c < nch) {
#pragma unroll
          for (int j = 0; j < 4; ++j) am = fmaxf(am, fmaxf(fabsf(bf16lo(v[j])), fabsf(bf16hi(v[j]))));
        }
        am = fmaxf(am, __shfl_xor(am, 1, 64));  // a block = 4 consecutive chunks = 4 consecutive lanes
        am = fmaxf(am, __shfl_xor(am, 2, 64));
        if (on) {
          const int byte = mx_scale_byte_nosat(am);
          const float inv = mx_inv_scale(byte);
          int w0 = 0, w1 = 0;
          w0 = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(bf16lo(v[0]) * inv), clamp448(bf16hi(v[0]) * inv), w0, false);
          w0 = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(bf16lo(v[1]) * inv), clamp448(bf16hi(v[1]) * inv), w0, true);
          w1 = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(bf16lo(v[2]) * inv), clamp448(bf16hi(v[2]) * inv), w1, false);
          w1 = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(bf16lo(v[3]) * inv), clamp448(bf16hi(v[3]) * inv), w1, true);
          const u32x2 ov = {(uint32_t)w0, (uint32_t)w1};
          *reinterpret_cast<u32x2*>(q8 + (size_t)(row0 + k) * ldq + c * 8) = ov;
          if ((lane & 3) == 0) reinterpret_cast<unsigned char*>(qscale)[mx_gemm_scale_offset(row0 + k, c >> 2, D >> 7)] = (unsigned char)byte;
        }
      }
    }
  }
  if (FP8 == 1) {
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const float am = wave_max(amax[k]);
      const float sc = am > 0.f ? am * (1.0f / 448.0f) : 1.0f;
      const float inv = 1.0f / sc;
      if (row0 + k >= M) continue;
      if (lane == 0) qscale[row0 + k] = sc;
      unsigned char* qr = q8 + (size_t)(row0 + k) * ldq;
#pragma unroll
      for (int i = 0; i < ROW_MAXC; ++i) {
        const int c = lane + 64 * i;
        if (FULL || c < nch) {
          const u32x4 v = raw[k][i];
          int w0 = 0, w1 = 0;
          w0 = __builtin_amdgcn_cvt_pk_fp8_f32(bf16lo(v[0]) * inv, bf16hi(v[0]) * inv, w0, false);
          w0 = __builtin_amdgcn_cvt_pk_fp8_f32(bf16lo(v[1]) * inv, bf16hi(v[1]) * inv, w0, true);
          w1 = __builtin_amdgcn_cvt_pk_fp8_f32(bf16lo(v[2]) * inv, bf16hi(v[2]) * inv, w1, false);
          w1 = __builtin_amdgcn_cvt_pk_fp8_f32(bf16lo(v[3]) * inv, bf16hi(v[3]) * inv, w1, true);
          u32x2 o = {(uint32_t)w0, (uint32_t)w1};
          *reinterpret_cast<u32x2*>(qr + c * 8) = o;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------
// RMSNorm(across heads) [+ RoPE], in place.  cs = [ntok][head_dim/2][2] fp32 (cos, sin)
// Rounding points follow diffusers RMSNorm under bf16: bf16(x * rstd) then bf16(. * w).
// RoPE is evaluated in fp32 on the bf16-rounded values (reference uses fp64: the two
// agree to ~1e-7 relative before the final bf16 rounding).
// ------------------------------------------------------------------------------------
template <bool FULL>
__global__ __launch_bounds__(256) void rmsnorm_rope_kernel(bf16* __restrict__ x0, const float* __restrict__ w0,
                                                           bf16* __restrict__ x1, const float* __restrict__ w1,
                                                           const float* __restrict__ cs, int M, int D, int ld,
                                                           int head_dim, float eps, int rope_rows) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  bf16* __restrict__ x = blockIdx.y ? x1 : x0;         // blockIdx.y selects the tensor (q / k of one fused qkv buffer)
  const float* __restrict__ w = blockIdx.y ? w1 : w0;
  const int cs_row = rope_rows > 0 ? row % rope_rows : row;  // RoPE table period when samples are stacked along M
  const int nch = D >> 3;
  bf16* xr = x + (size_t)row * ld;
  u32x4 raw[ROW_MAXC];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < ROW_MAXC; ++i) {
    const int c = lane + 64 * i;
    if (FULL || c < nch) {
      raw[i] = *reinterpret_cast<const u32x4*>(xr + c * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float v0 = bf16lo(raw[i][j]), v1 = bf16hi(raw[i][j]);
        s += v0 * v0 + v1 * v1;
      }
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(s) / (float)D + eps);
  const int half = head_dim >> 1;
  // (round 6) a lane's chunks are 64 chunks = 512 channels apart: whenever 512 is a multiple of head_dim (128 here) they all sit at the same
  // place of their head and share ONE cos / sin entry - two loads per row instead of twenty (the row's 512-byte table was read forty times over)
  const bool cs_once = cs != nullptr && (512 % head_dim) == 0;
  f32x4 csa = {0.f, 0.f, 0.f, 0.f}, csb = {0.f, 0.f, 0.f, 0.f};
  if (cs_once) {
    const float* p = cs + ((size_t)cs_row * half + (((lane * 8) % head_dim) >> 1)) * 2;
    csa = *reinterpret_cast<const f32x4*>(p);
    csb = *reinterpret_cast<const f32x4*>(p + 4);
  }
#pragma unroll
  for (int i = 0; i < ROW_MAXC; ++i) {
    const int c = lane + 64 * i;
    if (FULL || c < nch) {
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(w + c * 8), w1 = *reinterpret_cast<const f32x4*>(w + c * 8 + 4);
      u32x4 o;
      f32x4 cs0 = csa, cs1 = csb;
      if (cs != nullptr && !cs_once) {
        const int pair0 = ((c * 8) % head_dim) >> 1;  // first of the 4 (even,odd) pairs of this chunk
        const float* p = cs + ((size_t)cs_row * half + pair0) * 2;
        cs0 = *reinterpret_cast<const f32x4*>(p);
        cs1 = *reinterpret_cast<const f32x4*>(p + 4);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float ww0 = j < 2 ? w0[2 * j] : w1[2 * j - 4], ww1 = j < 2 ? w0[2 * j + 1] : w1[2 * j - 3];
        float v0 = round_bf16(round_bf16(bf16lo(raw[i][j]) * rstd) * ww0);
        float v1 = round_bf16(round_bf16(bf16hi(raw[i][j]) * rstd) * ww1);
        if (cs != nullptr) {
          const float co = j < 2 ? cs0[2 * j] : cs1[2 * j - 4], si = j < 2 ? cs0[2 * j + 1] : cs1[2 * j - 3];
          const float r0 = v0 * co - v1 * si, r1 = v0 * si + v1 * co;
          v0 = r0;
          v1 = r1;
        }
        o[j] = pack_bf16(v0, v1);
      }
      *reinterpret_cast<u32x4*>(xr + c * 8) = o;
    }
  }
}

// ------------------------------------------------------------------------------------
// Ulysses send-side layout transform fused into the RMSNorm(+RoPE) pass (reference design
// chronoedit_diffsynth/wan_video_new_chronoedit.py:330-355,1448-1453: q/k are normalised across ALL heads and rotated
// BEFORE the head split, so both are token-local).  For each of `nt` column blocks of x ([M][ldx], block i starts at
// column col_i and is D wide): RMSNorm * w_i (+ RoPE) when w_i != NULL, a plain copy otherwise (v), written straight into
// the all-to-all SEND buffer
//     send[dst rank r][row m][block i][Dl],   Dl = D / W,   column n of block i goes to r = n / Dl, offset n % Dl
// so that the chunk for rank r is contiguous (what all_to_all_single wants) and no permute().contiguous() pass exists.
// W == 1 degenerates to a packed [M][nt][D] copy.  Same rounding points as rmsnorm_rope_kernel.
// ------------------------------------------------------------------------------------
template <bool FULL>
__global__ __launch_bounds__(256) void rope_scatter_kernel(const bf16* __restrict__ x, int ldx, bf16* __restrict__ send,
                                                           int M, int D, int W, int nt, int col0, const float* __restrict__ w0,
                                                           int col1, const float* __restrict__ w1, int col2,
                                                           const float* __restrict__ w2, const float* __restrict__ cs,
                                                           int head_dim, float eps, int rope_rows) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int ti = blockIdx.y;
  const int col = ti == 0 ? col0 : ti == 1 ? col1 : col2;
  const float* __restrict__ w = ti == 0 ? w0 : ti == 1 ? w1 : w2;
  const int cs_row = rope_rows > 0 ? row % rope_rows : row;
  const int nch = D >> 3;
  const int Dl = D / W;
  const bf16* xr = x + (size_t)row * ldx + col;
  u32x4 raw[ROW_MAXC];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < ROW_MAXC; ++i) {
    const int c = lane + 64 * i;
    if (FULL || c < nch) {
      raw[i] = *reinterpret_cast<const u32x4*>(xr + c * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float v0 = bf16lo(raw[i][j]), v1 = bf16hi(raw[i][j]);
        s += v0 * v0 + v1 * v1;
      }
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(s) / (float)D + eps);
  const int half = head_dim >> 1;
#pragma unroll
  for (int i = 0; i < ROW_MAXC; ++i) {
    const int c = lane + 64 * i;
    if (FULL || c < nch) {
      u32x4 o = raw[i];
      if (w != nullptr) {
        const f32x4 wa = *reinterpret_cast<const f32x4*>(w + c * 8), wb = *reinterpret_cast<const f32x4*>(w + c * 8 + 4);
        f32x4 cs0, cs1;
        if (cs != nullptr) {
          const int pair0 = ((c * 8) % head_dim) >> 1;
          const float* p = cs + ((size_t)cs_row * half + pair0) * 2;
          cs0 = *reinterpret_cast<const f32x4*>(p);
          cs1 = *reinterpret_cast<const f32x4*>(p + 4);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float ww0 = j < 2 ? wa[2 * j] : wb[2 * j - 4], ww1 = j < 2 ? wa[2 * j + 1] : wb[2 * j - 3];
          float v0 = round_bf16(round_bf16(bf16lo(raw[i][j]) * rstd) * ww0);
          float v1 = round_bf16(round_bf16(bf16hi(raw[i][j]) * rstd) * ww1);
          if (cs != nullptr) {
            const float co = j < 2 ? cs0[2 * j] : cs1[2 * j - 4], si = j < 2 ? cs0[2 * j + 1] : cs1[2 * j - 3];
            const float r0 = v0 * co - v1 * si, r1 = v0 * si + v1 * co;
            v0 = r0;
            v1 = r1;
          }
          o[j] = pack_bf16(v0, v1);
        }
      }
      const int n = c * 8, r = n / Dl, cc = n - r * Dl;
      *reinterpret_cast<u32x4*>(send + (((size_t)r * M + row) * nt + ti) * Dl + cc) = o;
    }
  }
}

// ------------------------------------------------------------------------------------
// K2: timestep sinusoid + small GEMV chain (one wave per output feature)
// ------------------------------------------------------------------------------------
// out[i] = cos(t * f_i) for i < half, sin(t * f_i) otherwise;  f_i = exp(-ln(1e4) * i / half)
template <typename TT>
__global__ void timestep_sinusoid_kernel(const TT* __restrict__ t, float* __restrict__ out, int dim) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int half = dim >> 1;
  if (i >= dim) return;
  const int k = i < half ? i : i - half;
  const float freq = expf(-9.210340371976184f * (float)k / (float)half);
  const float arg = (float)t[0] * freq;
  out[i] = i < half ? cosf(arg) : sinf(arg);
}

// y[n] = post( W[n,:] . pre(x) + bias[n] );  WT = float or bf16.
// flags: bit0 pre: x <- bf16(silu(x));  bit1 post: silu;  bit2 post: round to bf16 (stored as fp32)
// One workgroup = GEMV_ROWS output features (four waves, GEMV_ROWS / 4 rows each); pre(x) is formed ONCE per workgroup in
// LDS (it used to be recomputed - one expf per element - by every row) and the weights stream in 16-byte loads (they used to
// come two bytes per lane per load: 1.07 TB/s on the 315 MB time_proj matrix).
constexpr int GEMV_ROWS = 16;
template <typename WT>
__global__ __launch_bounds__(256) void gemv_kernel(const WT* __restrict__ W, const float* __restrict__ x,
                                                   const float* __restrict__ bias, float* __restrict__ y, int N, int K,
                                                   int flags) {
  extern __shared__ __attribute__((aligned(16))) float xs[];
  for (int k = threadIdx.x; k < K; k += 256) {
    float xv = x[k];
    if (flags & 1) xv = round_bf16(silu(xv));
    xs[k] = xv;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int VEC = 16 / sizeof(WT);  // elements per 16-byte load: 8 bf16 or 4 fp32
  const bool vec = (K % (64 * VEC)) == 0 && ((size_t)W & 15) == 0;
  for (int r = 0; r < GEMV_ROWS / 4; ++r) {
    const int n = blockIdx.x * GEMV_ROWS + r * 4 + wave;
    if (n >= N) break;
    const WT* wr = W + (size_t)n * K;
    float acc = 0.f;
    if (vec) {
      for (int k = lane * VEC; k < K; k += 64 * VEC) {
        const u32x4 wv = *reinterpret_cast<const u32x4*>(wr + k);
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(xs + k);
        if (sizeof(WT) == 2) {
          const f32x4 x1 = *reinterpret_cast<const f32x4*>(xs + k + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            acc += bf16lo(wv[j]) * (j < 2 ? x0[2 * j] : x1[2 * j - 4]);
            acc += bf16hi(wv[j]) * (j < 2 ? x0[2 * j + 1] : x1[2 * j - 3]);
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) acc += __uint_as_float(wv[j]) * x0[j];
        }
      }
    } else {
      for (int k = lane; k < K; k += 64) acc += (float)wr[k] * xs[k];
    }
    acc = wave_sum(acc);
    if (lane == 0) {
      float rr = acc + (bias ? bias[n] : 0.f);
      if (flags & 4) rr = round_bf16(rr);
      if (flags & 2) rr = silu(rr);
      y[n] = rr;
    }
  }
}

// mod[l][j][d] = table[l][j][d] + v[j][d]  (+1 on the rows flagged in one_mask), fp32.
// Reference: (scale_shift_table + temb.float()).chunk(6) then (1 + scale), transformer_chronoedit.py:274-279,451.
__global__ void modulation_kernel(const float* __restrict__ table, const float* __restrict__ v, float* __restrict__ mod,
                                  int L, int J, int D, int v_rows, int one_mask) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)L * J * D;
  if (idx >= total) return;
  const int d = idx % D;
  const int j = (idx / D) % J;
  const float add = v[(size_t)(v_rows == 1 ? 0 : j) * D + d];
  float r = table[idx] + add;
  if ((one_mask >> j) & 1) r = 1.0f + r;
  mod[idx] = r;
}

// ------------------------------------------------------------------------------------
// K1 patchify (im2col of the k=s=(1,2,2) Conv3d) and K18 unpatchify
// x [C][T][H][W] bf16 -> cols [N = T*(H/2)*(W/2)][Kpad], k = c*4 + dh*2 + dw, zero padded
// ------------------------------------------------------------------------------------
// row0 / nrows: only token rows [row0, row0 + nrows) are produced (cols has nrows rows; rows past the last token are zero):
// the local shard of a sequence-parallel rank.
__global__ void patchify_kernel(const bf16* __restrict__ x, bf16* __restrict__ cols, int C, int T, int H, int W, int Kpad,
                                int row0, int nrows) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int h2 = H >> 1, w2 = W >> 1;
  const size_t ntok = (size_t)T * h2 * w2;
  const size_t total = (size_t)nrows * Kpad;
  if (idx >= total) return;
  const int k = idx % Kpad;
  const size_t tok = idx / Kpad + row0;
  bf16 v = (bf16)0.f;
  if (k < C * 4 && tok < ntok) {
    const int c = k >> 2, dh = (k >> 1) & 1, dw = k & 1;
    const int wq = tok % w2, hq = (tok / w2) % h2, t = tok / ((size_t)w2 * h2);
    v = x[(((size_t)c * T + t) * H + (hq * 2 + dh)) * W + wq * 2 + dw];
  }
  cols[idx] = v;
}

// y [N][ldy >= 4*Cout], col = (dh*2+dw)*Cout + c  ->  out [Cout][T][H][W]
// (reshape/permute at transformer_chronoedit.py:463-467)
__global__ void unpatchify_kernel(const bf16* __restrict__ y, bf16* __restrict__ out, int Cout, int T, int H, int W, int ldy) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)Cout * T * H * W;
  if (idx >= total) return;
  const int w = idx % W, h = (idx / W) % H, t = (idx / ((size_t)W * H)) % T, c = idx / ((size_t)W * H * T);
  const int h2 = H >> 1, w2 = W >> 1;
  const size_t tok = ((size_t)t * h2 + (h >> 1)) * w2 + (w >> 1);
  out[idx] = y[tok * ldy + ((h & 1) * 2 + (w & 1)) * Cout + c];
}

// ------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------
CE_API int ce_ln_affine_bf16(const void* x, void* y, const float* a, const float* b, int M, int D, int ldx, int ldy,
                                 float eps, int ab_rows, int ab_stride, hipStream_t stream) {
  if (!x || !y || !a || !b) return CE_ERR_ARG;
  if (M <= 0 || D <= 0 || (D & 7) || D > 64 * 8 * ROW_MAXC || (ldx & 7) || (ldy & 7) || (ab_rows > 0 && (ab_stride & 3))) return CE_ERR_SHAPE;
  // rows per wave (see the kernel): the full-width form shares each (a, b) chunk between CE_LN_ROWS rows of one sample
  const bool multi = D == 64 * 8 * ROW_MAXC && (ab_rows <= 0 || ab_rows % CE_LN_ROWS == 0);
  if (multi)
    hipLaunchKernelGGL((ln_affine_kernel<0, true, CE_LN_ROWS>), dim3((M + 4 * CE_LN_ROWS - 1) / (4 * CE_LN_ROWS)), dim3(256), 0, stream,
                       (const bf16*)x, (bf16*)y, a, b, M, D, ldx, ldy, eps, ab_rows, ab_stride, nullptr, nullptr, 0);
  else if (D == 64 * 8 * ROW_MAXC)
    hipLaunchKernelGGL((ln_affine_kernel<0, true, 1>), dim3((M + 3) / 4), dim3(256), 0, stream, (const bf16*)x, (bf16*)y, a, b, M,
                       D, ldx, ldy, eps, ab_rows, ab_stride, nullptr, nullptr, 0);
  else
    hipLaunchKernelGGL((ln_affine_kernel<0, false, 1>), dim3((M + 3) / 4), dim3(256), 0, stream, (const bf16*)x, (bf16*)y, a, b, M,
                       D, ldx, ldy, eps, ab_rows, ab_stride, nullptr, nullptr, 0);
  return (int)hipGetLastError();
}

CE_API int ce_ln_affine_fp8(const void* x, void* q, float* scale, const float* a, const float* b, int M, int D, int ldx, int ldq,
                                float eps, int ab_rows, int ab_stride, hipStream_t stream) {
  if (!x || !q || !scale || !a || !b) return CE_ERR_ARG;
  if (M <= 0 || D <= 0 || (D & 7) || D > 64 * 8 * ROW_MAXC || (ldx & 7) || (ldq & 7) || (ab_rows > 0 && (ab_stride & 3))) return CE_ERR_SHAPE;
  // (one row per wave here: with two the fp8 form needs 360 registers - the quantised rows stay live for the amax)
  if (D == 64 * 8 * ROW_MAXC)
    hipLaunchKernelGGL((ln_affine_kernel<1, true, 1>), dim3((M + 3) / 4), dim3(256), 0, stream, (const bf16*)x, nullptr, a, b, M, D,
                       ldx, 0, eps, ab_rows, ab_stride, (unsigned char*)q, scale, ldq);
  else
    hipLaunchKernelGGL((ln_affine_kernel<1, false, 1>), dim3((M + 3) / 4), dim3(256), 0, stream, (const bf16*)x, nullptr, a, b, M, D,
                       ldx, 0, eps, ab_rows, ab_stride, (unsigned char*)q, scale, ldq);
  return (int)hipGetLastError();
}

/* ce_ln_affine_bf16 followed by ce_quant_rows_mxfp8 in one pass (q: e4m3 bytes, scale8: E8M0 block scales in the tiled layout of
 * ce_gemm_mxfp8, ceil(M / 128) * (D / 128) * 512 bytes).  D % 128 == 0. */
CE_API int ce_ln_affine_mxfp8(const void* x, void* q, void* scale8, const float* a, const float* b, int M, int D, int ldx, int ldq, float eps,
                                  int ab_rows, int ab_stride, hipStream_t stream) {
  if (!x || !q || !scale8 || !a || !b) return CE_ERR_ARG;
  if (M <= 0 || D <= 0 || (D & 127) || D > 64 * 8 * ROW_MAXC || (ldx & 7) || (ldq & 7) || (ab_rows > 0 && (ab_stride & 3))) return CE_ERR_SHAPE;
  if (D == 64 * 8 * ROW_MAXC)
    hipLaunchKernelGGL((ln_affine_kernel<2, true, 1>), dim3((M + 3) / 4), dim3(256), 0, stream, (const bf16*)x, nullptr, a, b, M, D, ldx, 0, eps,
                       ab_rows, ab_stride, (unsigned char*)q, (float*)scale8, ldq);
  else
    hipLaunchKernelGGL((ln_affine_kernel<2, false, 1>), dim3((M + 3) / 4), dim3(256), 0, stream, (const bf16*)x, nullptr, a, b, M, D, ldx, 0, eps,
                       ab_rows, ab_stride, (unsigned char*)q, (float*)scale8, ldq);
  return (int)hipGetLastError();
}

CE_API int ce_rmsnorm_rope_bf16(void* x, const float* w, void* x2, const float* w2, const float* cos_sin, int M, int D, int ld,
                                    int head_dim, float eps, int rope_rows, hipStream_t stream) {
  if (!x || !w || (x2 && !w2)) return CE_ERR_ARG;
  if (M <= 0 || D <= 0 || (D & 7) || D > 64 * 8 * ROW_MAXC || (ld & 7) || (head_dim & 7) || D % head_dim) return CE_ERR_SHAPE;
  if (D == 64 * 8 * ROW_MAXC)
    hipLaunchKernelGGL(rmsnorm_rope_kernel<true>, dim3((M + 3) / 4, x2 ? 2 : 1), dim3(256), 0, stream, (bf16*)x, w, (bf16*)x2, w2,
                       cos_sin, M, D, ld, head_dim, eps, rope_rows);
  else
    hipLaunchKernelGGL(rmsnorm_rope_kernel<false>, dim3((M + 3) / 4, x2 ? 2 : 1), dim3(256), 0, stream, (bf16*)x, w, (bf16*)x2, w2,
                       cos_sin, M, D, ld, head_dim, eps, rope_rows);
  return (int)hipGetLastError();
}

CE_API int ce_timestep_sinusoid(const int64_t* t, float* out, int dim, hipStream_t stream) {
  if (!t || !out || dim <= 0 || (dim & 1)) return CE_ERR_ARG;
  hipLaunchKernelGGL(timestep_sinusoid_kernel<int64_t>, dim3((dim + 255) / 256), dim3(256), 0, stream, t, out, dim);
  return (int)hipGetLastError();
}

CE_API int ce_timestep_sinusoid_f32(const float* t, float* out, int dim, hipStream_t stream) {
  if (!t || !out || dim <= 0 || (dim & 1)) return CE_ERR_ARG;
  hipLaunchKernelGGL(timestep_sinusoid_kernel<float>, dim3((dim + 255) / 256), dim3(256), 0, stream, t, out, dim);
  return (int)hipGetLastError();
}

CE_API int ce_gemv(const void* W, int w_is_bf16, const float* x, const float* bias, float* y, int N, int K, int flags,
                       hipStream_t stream) {
  if (!W || !x || !y || N <= 0 || K <= 0) return CE_ERR_ARG;
  if (K > 16384) return CE_ERR_SHAPE;  // pre(x) is staged in LDS (64 KiB)
  if (w_is_bf16)
    hipLaunchKernelGGL(gemv_kernel<bf16>, dim3((N + GEMV_ROWS - 1) / GEMV_ROWS), dim3(256), (size_t)K * 4, stream, (const bf16*)W, x, bias, y,
                       N, K, flags);
  else
    hipLaunchKernelGGL(gemv_kernel<float>, dim3((N + GEMV_ROWS - 1) / GEMV_ROWS), dim3(256), (size_t)K * 4, stream, (const float*)W, x, bias,
                       y, N, K, flags);
  return (int)hipGetLastError();
}

CE_API int ce_modulation(const float* table, const float* v, float* mod, int L, int J, int D, int v_rows, int one_mask,
                             hipStream_t stream) {
  if (!table || !v || !mod || L <= 0 || J <= 0 || D <= 0 || (v_rows != 1 && v_rows != J)) return CE_ERR_ARG;
  const size_t total = (size_t)L * J * D;
  hipLaunchKernelGGL(modulation_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, table, v, mod, L, J, D,
                     v_rows, one_mask);
  return (int)hipGetLastError();
}

CE_API int ce_patchify_rows_bf16(const void* x, void* cols, int C, int T, int H, int W, int Kpad, int row0, int nrows,
                                     hipStream_t stream) {
  if (!x || !cols || (H & 1) || (W & 1) || Kpad < C * 4 || row0 < 0 || nrows <= 0) return CE_ERR_ARG;
  const size_t total = (size_t)nrows * Kpad;
  hipLaunchKernelGGL(patchify_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const bf16*)x, (bf16*)cols,
                     C, T, H, W, Kpad, row0, nrows);
  return (int)hipGetLastError();
}

CE_API int ce_patchify_bf16(const void* x, void* cols, int C, int T, int H, int W, int Kpad, hipStream_t stream) {
  if ((H & 1) || (W & 1)) return CE_ERR_ARG;
  return ce_patchify_rows_bf16(x, cols, C, T, H, W, Kpad, 0, T * (H / 2) * (W / 2), stream);
}

CE_API int ce_rope_scatter_bf16(const void* x, int ldx, void* send, int M, int D, int W, int nt, int col0, const float* w0,
                                    int col1, const float* w1, int col2, const float* w2, const float* cos_sin, int head_dim,
                                    float eps, int rope_rows, hipStream_t stream) {
  if (!x || !send || nt < 1 || nt > 3 || W < 1) return CE_ERR_ARG;
  if (M <= 0 || D <= 0 || (D & 7) || D > 64 * 8 * ROW_MAXC || (ldx & 7) || (head_dim & 7) || D % head_dim || D % W ||
      ((D / W) & 7) || (col0 & 7) || (col1 & 7) || (col2 & 7))
    return CE_ERR_SHAPE;
  if (D == 64 * 8 * ROW_MAXC)
    hipLaunchKernelGGL(rope_scatter_kernel<true>, dim3((M + 3) / 4, nt), dim3(256), 0, stream, (const bf16*)x, ldx, (bf16*)send, M,
                       D, W, nt, col0, w0, col1, w1, col2, w2, cos_sin, head_dim, eps, rope_rows);
  else
    hipLaunchKernelGGL(rope_scatter_kernel<false>, dim3((M + 3) / 4, nt), dim3(256), 0, stream, (const bf16*)x, ldx, (bf16*)send, M,
                       D, W, nt, col0, w0, col1, w1, col2, w2, cos_sin, head_dim, eps, rope_rows);
  return (int)hipGetLastError();
}

CE_API int ce_unpatchify_bf16(const void* y, void* out, int Cout, int T, int H, int W, int ldy, hipStream_t stream) {
  if (!y || !out || (H & 1) || (W & 1) || ldy < 4 * Cout) return CE_ERR_ARG;
  const size_t total = (size_t)Cout * T * H * W;
  hipLaunchKernelGGL(unpatchify_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const bf16*)y, (bf16*)out,
                     Cout, T, H, W, ldy);
  return (int)hipGetLastError();
}
