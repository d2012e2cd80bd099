// Small kernels of the conditioning encoders that run once per edit, outside the denoising loop
// (reference call sites: chronoedit_diffusers/pipeline_chronoedit.py:205-254; the arithmetic lives in the un-vendored
// transformers==4.57.1 CLIPVisionModel / UMT5EncoderModel, restated in oracle/clip_oracle.py and oracle/umt5_oracle.py):
//   ce_im2col_patch2d   CLIP patch embedding: Conv2d(k = s = P, no bias) as im2col + ce_gemm_bf16
//   ce_gather_rows      UMT5 token embedding lookup
//   ce_rmsnorm          T5LayerNorm: x * rsqrt(mean(x^2) + eps) in fp32, rounded to bf16, times the bf16 weight
//   ce_softmax_t5       softmax(scores + relative-position bias + key padding mask) in fp32 -> bf16 probabilities
// All HBM-bound and tiny next to the encoder GEMMs; one wave64 per row where rows exist.
#include <algorithm>

#include "ce_common.h"

namespace {

constexpr float NEG_BIG_F = -1.0e30f;

// cols[(b*gh + py)*gw + px][c*P*P + y*P + x] = img[b][c][py*P + y][px*P + x]; columns >= C*P*P are zero
__global__ __launch_bounds__(256) void im2col_patch2d_kernel(const bf16* __restrict__ img, bf16* __restrict__ cols, int B, int C, int H,
                                                             int W, int P, int Kpad) {
  const int gh = H / P, gw = W / P;
  const long long total = (long long)B * gh * gw * Kpad;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(i % Kpad);
    const long long row = i / Kpad;
    bf16 v = (bf16)0.f;
    if (k < C * P * P) {
      const int px = (int)(row % gw), py = (int)((row / gw) % gh), b = (int)(row / ((long long)gw * gh));
      const int c = k / (P * P), y = (k / P) % P, x = k % P;
      v = img[(((size_t)b * C + c) * H + (size_t)py * P + y) * W + (size_t)px * P + x];
    }
    cols[i] = v;
  }
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const bf16* __restrict__ table, const long long* __restrict__ ids,
                                                          bf16* __restrict__ out, int n, int D, int ldt, int ldo, int vocab) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  long long id = ids[row];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);  // out-of-range ids cannot fault (the tokenizer never emits them)
  const bf16* src = table + (size_t)id * ldt;
  bf16* dst = out + (size_t)row * ldo;
  for (int c = lane; c < (D >> 3); c += 64) *reinterpret_cast<u32x4*>(dst + c * 8) = *reinterpret_cast<const u32x4*>(src + c * 8);
}

constexpr int RMS_MAXC = 10;  // D <= 5120
__global__ __launch_bounds__(256) void rmsnorm_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, const bf16* __restrict__ w, int M,
                                                      int D, int ldx, int ldy, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nch = D >> 3;
  const bf16* xr = x + (size_t)row * ldx;
  u32x4 raw[RMS_MAXC];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < RMS_MAXC; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
      raw[i] = *reinterpret_cast<const u32x4*>(xr + c * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float a = bf16lo(raw[i][j]), b = bf16hi(raw[i][j]);
        ss += a * a + b * b;
      }
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)D + eps);
  bf16* yr = y + (size_t)row * ldy;
#pragma unroll
  for (int i = 0; i < RMS_MAXC; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
      const u32x4 wv = *reinterpret_cast<const u32x4*>(w + c * 8);
      u32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // normalised value rounded to bf16 first, then the bf16 x bf16 product rounded again (UMT5LayerNorm.forward)
        const float n0 = round_bf16(bf16lo(raw[i][j]) * rstd), n1 = round_bf16(bf16hi(raw[i][j]) * rstd);
        o[j] = pack_bf16(n0 * bf16lo(wv[j]), n1 * bf16hi(wv[j]));
      }
      *reinterpret_cast<u32x4*>(yr + c * 8) = o;
    }
  }
}

// one wave per (batch b, head h, query q) row of Lk scores: p = softmax(s + table[bucket(k - q)][h], keys >= valid[b] out)
constexpr int SM_MAXK = 16;  // Lk <= 64 * 16 = 1024
__global__ __launch_bounds__(256) void softmax_t5_kernel(const float* __restrict__ scores, bf16* __restrict__ probs, int rows, int heads,
                                                         int Lq, int Lk, int ld, int ldp, const int* __restrict__ bucket_lut,
                                                         const float* __restrict__ table, const int* __restrict__ valid) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int q = row % Lq, h = (row / Lq) % heads, b = row / (Lq * heads);
  const int nk = valid ? min(valid[b], Lk) : Lk;
  const float* sr = scores + (size_t)row * ld;
  float v[SM_MAXK];
  float mx = NEG_BIG_F;
#pragma unroll
  for (int i = 0; i < SM_MAXK; ++i) {
    const int k = lane + 64 * i;
    v[i] = NEG_BIG_F;
    if (k < nk) {
      float s = sr[k];
      if (table) s += table[bucket_lut[k - q + Lq - 1] * heads + h];
      v[i] = s;
      mx = fmaxf(mx, s);
    }
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < SM_MAXK; ++i) {
    const int k = lane + 64 * i;
    v[i] = k < nk ? __expf(v[i] - mx) : 0.f;
    sum += v[i];
  }
  const float inv = 1.0f / wave_sum(sum);
  bf16* pr = probs + (size_t)row * ldp;
#pragma unroll
  for (int i = 0; i < SM_MAXK; ++i) {
    const int k = lane + 64 * i;
    if (k < ldp) pr[k] = (bf16)(k < Lk ? v[i] * inv : 0.f);  // padding columns (GEMM K alignment) are written as zeros
  }
}

}  // namespace

CE_API int ce_im2col_patch2d_bf16(const void* img, void* cols, int B, int C, int H, int W, int P, int Kpad, hipStream_t stream) {
  if (!img || !cols) return CE_ERR_ARG;
  if (B <= 0 || C <= 0 || P <= 0 || H % P || W % P || Kpad < C * P * P) return CE_ERR_SHAPE;
  const long long total = (long long)B * (H / P) * (W / P) * Kpad;
  const int blocks = (int)std::min<long long>((total + 255) / 256, 65535);
  hipLaunchKernelGGL(im2col_patch2d_kernel, dim3(blocks), dim3(256), 0, stream, (const bf16*)img, (bf16*)cols, B, C, H, W, P, Kpad);
  return (int)hipGetLastError();
}

CE_API int ce_gather_rows_bf16(const void* table, const long long* ids, void* out, int n, int D, int ldt, int ldo, int vocab,
                                   hipStream_t stream) {
  if (!table || !ids || !out) return CE_ERR_ARG;
  if (n <= 0 || D <= 0 || vocab <= 0) return CE_ERR_SHAPE;
  if ((D & 7) || (ldt & 7) || (ldo & 7)) return CE_ERR_ALIGN;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, (const bf16*)table, ids, (bf16*)out, n, D, ldt, ldo,
                     vocab);
  return (int)hipGetLastError();
}

CE_API int ce_rmsnorm_bf16(const void* x, void* y, const void* w, int M, int D, int ldx, int ldy, float eps, hipStream_t stream) {
  if (!x || !y || !w) return CE_ERR_ARG;
  if (M <= 0 || D <= 0 || D > 64 * 8 * RMS_MAXC) return CE_ERR_SHAPE;
  if ((D & 7) || (ldx & 7) || (ldy & 7)) return CE_ERR_ALIGN;
  hipLaunchKernelGGL(rmsnorm_kernel, dim3((M + 3) / 4), dim3(256), 0, stream, (const bf16*)x, (bf16*)y, (const bf16*)w, M, D, ldx, ldy, eps);
  return (int)hipGetLastError();
}

CE_API int ce_softmax_t5_bf16(const float* scores, void* probs, int batch, int heads, int Lq, int Lk, int ld, int ldp,
                                  const int* bucket_lut, const float* table, const int* valid_len, hipStream_t stream) {
  if (!scores || !probs || ((table != nullptr) != (bucket_lut != nullptr))) return CE_ERR_ARG;
  if (batch <= 0 || heads <= 0 || Lq <= 0 || Lk <= 0 || Lk > 64 * SM_MAXK || ldp < Lk || ldp > 64 * SM_MAXK || ld < Lk) return CE_ERR_SHAPE;
  const int rows = batch * heads * Lq;
  hipLaunchKernelGGL(softmax_t5_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, scores, (bf16*)probs, rows, heads, Lq, Lk, ld, ldp,
                     bucket_lut, table, valid_len);
  return (int)hipGetLastError();
}
