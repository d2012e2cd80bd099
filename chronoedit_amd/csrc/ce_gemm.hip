// bf16 GEMM with fused epilogues for the DiT projections (K1,K3,K6,K10,K12,K13,K15,K16,K18):
//     C[M,N] = epilogue( A[M,K] . W[N,K]^T + bias[N] )
// A = activations (token-major, K contiguous), W = nn.Linear weight [out,in] (K contiguous): both
// MFMA operands are K-contiguous in memory, so every lane fetches its 8-element fragment as
// one 16-B access.  Reference call sites: transformer_chronoedit.py:58-60,106 (attention
// projections), diffusers FeedForward (:262,292), patch_embedding (:429), proj_out (:461).
//
// Bound: bf16 MFMA (dense contraction); algorithmic flops = 2*M*N*K per launch.
//
// v1 kernel: 128x128x64 block tile, 4 waves (2x2), each wave a 64x64 sub-tile as 4x4
// v_mfma_f32_16x16x32_bf16 accumulators; A/W tiles double-buffered in LDS with a 16-B-chunk XOR
// swizzle (conflict-free ds_read_b128 fragment reads); the next K-tile is prefetched
// global->VGPR under the MFMAs of the current one (split issue-early/write-late staging).
// The epilogue stages bf16(acc+bias) through LDS so residual reads and stores are 16 B/lane.
//
// Epilogues (rounding points mirror the reference's eager bf16 path):
//   EPI_BIAS       C = bf16(acc + bias)
//   EPI_BIAS_GELU  C = bf16(gelu_tanh(bf16(acc + bias)))                      (FFN up, K16)
//   EPI_BIAS_GELU_ERF  same with the exact (erf) GELU                         (image MLP, K3)
//   EPI_GATE_RES   C = bf16(float(res) + float(bf16(acc + bias)) * gate[n])   (K10/K15/K17;
//                  gate == nullptr -> 1.0, i.e. the plain bf16 residual add of :286)
#include <cmath>

#include "ce_common.h"

#define EPI_BIAS 0
#define EPI_BIAS_GELU 1
#define EPI_GATE_RES 2
#define EPI_BIAS_GELU_ERF 3
#define EPI_F32 4  // C is float*: raw fp32 accumulators (attention scores of the VAE mid block)
#define EPI_BIAS_ROW 6  // C = bf16(acc + bias[m]): bias along the rows of C - a product with swapped operand roles (V^T = W_v.X^T)
#define EPI_BIAS_T 7    // the transpose of the product is stored (C is [N][ldc]): ce_gemm_epi.h / ce_gemm384.hip
#define EPI_MUL 5  // C = bf16(bf16(acc + bias) * res): the gated activation of the UMT5 feed-forward (wi_1(x) * gelu(wi_0(x)))

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand tile
constexpr int LDC_BYTES = BN * 2 + 16;   // padded C staging row (272 B, 16-B aligned)

__device__ __forceinline__ int swz_chunk(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

template <int EPI>
__global__ __launch_bounds__(256) void gemm_bf16_128(const bf16* __restrict__ A, const bf16* __restrict__ W,
                                                     bf16* __restrict__ C, const float* __restrict__ bias,
                                                     const float* __restrict__ gate, const bf16* __restrict__ res, int M,
                                                     int N, int K, int lda, int ldw, int ldc, int ldres, int gate_rows,
                                                     int tiles_m, int tiles_n, int batch0, long long sA0, long long sA1,
                                                     long long sW0, long long sW1, long long sC0, long long sC1,
                                                     int a_seg_tiles, long long a_seg_extra) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 2 * TILE_BYTES];  // [buf][A|W]
  if (gridDim.y > 1) {  // strided batch (two levels, e.g. head within sample): element strides, C in units of its own type
    const int z0 = blockIdx.y % batch0, z1 = blockIdx.y / batch0;
    A += z0 * sA0 + z1 * sA1;
    W += z0 * sW0 + z1 * sW1;
    const long long oc = z0 * sC0 + z1 * sC1;
    C = EPI == EPI_F32 ? reinterpret_cast<bf16*>(reinterpret_cast<float*>(C) + oc) : C + oc;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // tile order: XCD-contiguous chunks, grouped 8 m-tiles x all n-tiles for L2 reuse of A rows
  const int nwg = tiles_m * tiles_n;
  const int wg = xcd_remap(blockIdx.x, nwg);
  constexpr int GROUP = 8;
  const int group_sz = GROUP * tiles_n;
  const int gid = wg / group_sz;
  const int first_m = gid * GROUP;
  const int gm = min(tiles_m - first_m, GROUP);
  const int tm = first_m + (wg % group_sz) % gm;
  const int tn = (wg % group_sz) / gm;
  const int m0 = tm * BM, n0 = tn * BN;

  // staging map: thread -> 4 rows (stride 32) x one 16-B chunk of the 64-wide K slab
  const int st_row = tid >> 3, st_ck = tid & 7;
  const bf16* a_src[4];
  const bf16* w_src[4];
  int st_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = st_row + 32 * i;
    a_src[i] = A + (size_t)min(m0 + r, M - 1) * lda + st_ck * 8;
    w_src[i] = W + (size_t)min(n0 + r, N - 1) * ldw + st_ck * 8;
    st_off[i] = r * (BK * 2) + (swz_chunk(r, st_ck) << 4);
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 ra[4], rw[4];
  const int KT = K / BK;

#pragma unroll
  for (int i = 0; i < 4; ++i) {
    ra[i] = *reinterpret_cast<const u32x4*>(a_src[i]);
    rw[i] = *reinterpret_cast<const u32x4*>(w_src[i]);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    *reinterpret_cast<u32x4*>(smem + st_off[i]) = ra[i];
    *reinterpret_cast<u32x4*>(smem + TILE_BYTES + st_off[i]) = rw[i];
  }
  __syncthreads();

  const int fr = lane & 15, fg = lane >> 4;
  for (int kt = 0; kt < KT; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < KT) {
      const int koff = (kt + 1) * BK;
      // segmented A (ce_gemm_aseg_bf16): every a_seg_tiles K-tiles the source jumps a_seg_extra elements further
      const long long koff_a = koff + (a_seg_tiles > 0 ? ((kt + 1) / a_seg_tiles) * a_seg_extra : 0ll);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ra[i] = *reinterpret_cast<const u32x4*>(a_src[i] + koff_a);
        rw[i] = *reinterpret_cast<const u32x4*>(w_src[i] + koff);
      }
    }
    const unsigned char* sA = smem + cur * 2 * TILE_BYTES;
    const unsigned char* sW = sA + TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], wf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = wm * 64 + i * 16 + fr;
        af[i] = *reinterpret_cast<const bf16x8*>(sA + r * (BK * 2) + (swz_chunk(r, fg + 4 * ks) << 4));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = wn * 64 + j * 16 + fr;
        wf[j] = *reinterpret_cast<const bf16x8*>(sW + r * (BK * 2) + (swz_chunk(r, fg + 4 * ks) << 4));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], wf[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < KT) {
      unsigned char* dA = smem + (cur ^ 1) * 2 * TILE_BYTES;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        *reinterpret_cast<u32x4*>(dA + st_off[i]) = ra[i];
        *reinterpret_cast<u32x4*>(dA + TILE_BYTES + st_off[i]) = rw[i];
      }
    }
    __syncthreads();
  }

  // C fragment layout (16x16x32): col = lane & 15, row = 4 * (lane >> 4) + reg
  if (EPI == EPI_F32) {  // fp32 result straight from the accumulators (small GEMMs only)
    float* Cf = reinterpret_cast<float*>(C);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + fr;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = m0 + wm * 64 + i * 16 + fg * 4 + r;
          if (m < M && n < N) Cf[(size_t)m * ldc + n] = acc[i][j][r];
        }
    }
    return;
  }
  // ---- epilogue: bf16(acc + bias) -> LDS (row-major, padded) -> 16-B row-contiguous stores
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int cl = wn * 64 + j * 16 + fr;
    const int n = n0 + cl;
    const float bv = (EPI != EPI_BIAS_ROW && bias != nullptr && n < N) ? bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rl = wm * 64 + i * 16 + fg * 4 + r;
        const float br = EPI == EPI_BIAS_ROW ? bias[min(m0 + rl, M - 1)] : bv;
        *reinterpret_cast<bf16*>(smem + rl * LDC_BYTES + cl * 2) = (bf16)(acc[i][j][r] + br);
      }
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int c = tid + 256 * t;
    const int rl = c >> 4, cc = c & 15;
    const int m = m0 + rl, n = n0 + cc * 8;
    if (m < M && n < N) {
      u32x4 v = *reinterpret_cast<const u32x4*>(smem + rl * LDC_BYTES + cc * 16);
      if (EPI == EPI_BIAS_GELU) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = pack_bf16(gelu_tanh(bf16lo(v[q])), gelu_tanh(bf16hi(v[q])));
      } else if (EPI == EPI_BIAS_GELU_ERF) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = pack_bf16(gelu_erf(bf16lo(v[q])), gelu_erf(bf16hi(v[q])));
      } else if (EPI == EPI_MUL) {
        const u32x4 rv = *reinterpret_cast<const u32x4*>(res + (size_t)m * ldres + n);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = pack_bf16(bf16lo(rv[q]) * bf16lo(v[q]), bf16hi(rv[q]) * bf16hi(v[q]));
      } else if (EPI == EPI_GATE_RES) {
        const u32x4 rv = *reinterpret_cast<const u32x4*>(res + (size_t)m * ldres + n);
        float g[8];
        if (gate != nullptr) {
          const float* gp = gate + (gate_rows > 0 ? (size_t)(m / gate_rows) * N : 0) + n;  // per-sample gate rows
          const f32x4 g0 = *reinterpret_cast<const f32x4*>(gp), g1 = *reinterpret_cast<const f32x4*>(gp + 4);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            g[q] = g0[q];
            g[4 + q] = g1[q];
          }
        } else {
#pragma unroll
          for (int q = 0; q < 8; ++q) g[q] = 1.0f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
          v[q] = pack_bf16(mul_then_add(bf16lo(v[q]), g[2 * q], bf16lo(rv[q])), mul_then_add(bf16hi(v[q]), g[2 * q + 1], bf16hi(rv[q])));
      }
      *reinterpret_cast<u32x4*>(C + (size_t)m * ldc + n) = v;
    }
  }
}

}  // namespace

extern "C" int ce_gemm256_supported(int M, int N, int K, int lda, int ldw);
extern "C" int ce_gemm256_launch(const void* A, const void* W, void* C, const float* bias, int epilogue, const float* gate,
                                 const void* res, int M, int N, int K, int lda, int ldw, int ldc, int ldres, int gate_rows,
                                 int a_seg_k, long long a_seg_stride, int w_seg_k, long long w_seg_stride, hipStream_t stream);

extern "C" void ce_gemm256_set_staggered(int on);
extern "C" int ce_gemm384_launch(const void* A, const void* W, void* C, const float* bias, int epilogue, const float* gate,
                                 const void* res, int M, int N, int K, int lda, int ldw, int ldc, int ldres, int gate_rows,
                                 int a_seg_k, long long a_seg_stride, int w_seg_k, long long w_seg_stride, hipStream_t stream);
extern "C" int ce_gemm288_launch(const void* A, const void* W, void* C, const float* bias, int epilogue, const float* gate,
                                 const void* res, int M, int N, int K, int lda, int ldw, int ldc, int ldres, int gate_rows,
                                 int a_seg_k, long long a_seg_stride, int w_seg_k, long long w_seg_stride, hipStream_t stream);
extern "C" int ce_gemm256w4_launch(const void* A, const void* W, void* C, const float* bias, int epilogue, const float* gate,
                                   const void* res, int M, int N, int K, int lda, int ldw, int ldc, int ldres, int gate_rows,
                                   int a_seg_k, long long a_seg_stride, int w_seg_k, long long w_seg_stride, int nsa, hipStream_t stream);

// kernel selection: -1 = automatic (for large shapes the one-wave-per-SIMD LDS-DMA kernels, macro tile 384 x 256 or 256 x 256 by
// the round count of the shape: ce_gemm_bf16_tile_rows), 0 = always the
// 128-tile kernel; whenever the shape allows the 256-tile kernel: 1 = its 8-wave / 8-phase main loop, 2 = the same staggered (two
// wave groups one barrier apart), 3 / 4 = the one-wave-per-SIMD main loop of ce_gemm256w4.hip with an A ring of 3 / 2 stages, 5 = 3 stages and one barrier per K-tile,
// 6 = the 384 x 256 macro tile of ce_gemm384.hip (4 waves, 192 x 128 wave tiles, one barrier per K-tile), 7 = its 288 x 256 form (144 x 128 wave tiles)
CE_KNOB g_gemm_variant = -1;
#ifdef CE_DIAGNOSTICS
CE_API int ce_set_gemm_variant(int v) {
  const int old = g_gemm_variant;
  g_gemm_variant = v;
  ce_gemm256_set_staggered(v == 2);
  return old;
}
#endif

// Column k of A lives at A + (k / a_seg_k) * a_seg_stride + m * lda + k % a_seg_k: the layout an all-to-all leaves the
// attention output in ([source rank][local row][D / W], chronoedit_amd/parallel.py), consumed by the out-projection
// without a gather pass.  a_seg_k == 0 (or >= K): plain row-major A.
// ... and the same for W (w_seg_k, w_seg_stride): weights re-packed K-slab-major ([K/64][N][64]) so that every 16 KiB
// half-tile of the LDS-DMA stream is one contiguous block (tools/probes/l2_pattern_probe.hip: 21.7 vs 18.3 TB/s for the
// row-strided form).  Segmented W needs the 256-tile kernel (CE_ERR_SHAPE otherwise).
extern "C" void ce_gemm256_workspace(hipStream_t stream, float** ws, size_t* bytes, int* cus);

// Which macro tile for a large GEMM: 384 x 256 (ce_gemm384.hip) or 256 x 256 (ce_gemm256w4.hip)?  Their main loops run at the same
// rate per flop within 1 % (profiles/r03_gemm_variants_ab.txt); what differs is how the tile count falls on the 256 CUs.  Cost model:
// full rounds of workgroups, plus the last partial round - a whole round if it cannot be cut along K, 1 / split + an eighth of a
// round (slab write, reduce launch) if it can - times the tile area.  M = 14400 at N = 5120: 1140 tiles of 256 x 256 = 4 rounds + a
// half round + reduce against 760 tiles of 384 x 256 = 2.97 rounds (measured +3 ... +4 %); N = 13824: 12.02 against 8.02 rounds
// (256 x 256 wins by 2.6 %); the V^T product (M = 5120 = 13.3 tiles of 384 rows: 5 % of padding) stays on 256 x 256.
// Exported as a pure function of the shape, the CU count and the split-K workspace size (tests/test_host_cpu.py pins the step's choices).
CE_API int ce_gemm_bf16_tile_rows(int M, int N, int K, int cus, long long ws_bytes) {
  if (M <= 0 || N <= 0 || K < 64 || cus <= 0) return 0;
  const int kt = K / 64;
  // Round 6: refitted on the 41 (shape, tile) timings of profiles/r06_gemm_tile_choice.txt - the five large GEMMs of a block at every row count
  // the engine runs them on: M = 7 200 (distilled B = 1 step, BASELINE configs[2]), 14 400 (configs[1]), 13 068 / 26 136 (configs[4]), 28 800 /
  // 57 600 (configs[3] on one GPU), 3 648 / 7 296 (a rank of the 8-GPU split); the round-3 form was fitted at M = 14 400 only and left 2.1 ms
  // per forward on the table at M = 7 200.  In microseconds: a K-tile of a 256 x 256 tile costs 1.40 us on a busy chip, of a 384 x 256 tile
  // 1.5 x that x 0.96 (a sixth fewer operand bytes per flop), x 1.04 when the A operand does not stay in the 256 MB last-level cache beside W
  // (M K 2 B > 128 MiB: the M = 14 400 FFN-up runs the large tile 2.3 % slower, the M = 7 200 one 3.0 % faster).  A partial last round that
  // cannot be cut along K costs fill^0.3 of a round (measured 0.6 ... 0.8 of a round at 48 ... 59 % fill: fewer tiles share the L2 and the
  // power budget); one that can costs its K share plus the fp32 slabs, written and read back at ~6 TB/s (the round-3 form charged an eighth of
  // a round whatever K and the slab volume were).  On the 41 shapes the picks lose 0.4 % of the summed best-tile time (round-3 form: 2.3 %).
  auto cost = [&](int bm) -> double {
    const long long nwg = (long long)((M + bm - 1) / bm) * ((N + 255) / 256);
    const long long full = nwg / cus;
    const int tail = (int)(nwg % cus);
    // (288 x 256, round 6: 1.125 x the area of the 256-row tile, a twentieth fewer operand bytes per flop; priced between the other two)
    const double llc = (double)M * K * 2.0 > 128.0 * 1048576.0 ? 1.04 : 1.0;
    const double c = bm == 256 ? 1.40 : bm == 288 ? 1.40 * 1.125 * 0.985 * llc : 1.40 * 1.5 * 0.96 * llc;
    double t = (double)full * kt * c;
    if (tail > 0) {
      int split = 1;
      for (int sp = cus / tail < 8 ? cus / tail : 8; sp >= 2; --sp)
        if (kt % (2 * sp) == 0 && (long long)tail * sp * bm * 256 * (long long)sizeof(float) <= ws_bytes) {
          split = sp;
          break;
        }
      if (split > 1) t += (double)kt / split * c + (double)tail * split * bm * 256.0 * 8.0 / 6.0e6;
      else t += (double)kt * c * pow((double)tail / cus, 0.3);
    }
    return t;
  };
  const double c384 = cost(384), c256 = cost(256);
  // The 288-row form of the 384-row kernel (round 6) only where all three timings of profiles/r06_gemm_tile_choice_3tiles.txt agree it wins:
  // M a whole number of 288-row tiles and a last round that is full enough to run whole (no slabs) - M = 7 200 at N = K = 5120 (the
  // out-projections and the cross-attention's q of the distilled B = 1 step): 500 workgroups = 1.95 rounds against 380 = 1.48 rounds whose tail
  // goes through 97 MB of fp32 slabs: 0.267 against 0.314 ms.  Elsewhere its main loop is 2-6 % behind the larger tile's.
  if (M % 288 == 0) {
    const long long nwg = (long long)(M / 288) * ((N + 255) / 256);
    const int tail = (int)(nwg % cus);
    if (tail == 0 || tail * 10 >= cus * 9) {
      const double c288 = cost(288);
      if (c288 < 0.97 * (c384 < c256 ? c384 : c256)) return 288;
    }
  }
  return c384 <= c256 ? 384 : 256;
}

static int auto_tile_rows(int M, int N, int K, hipStream_t stream) {
  float* ws = nullptr;
  size_t ws_bytes = 0;
  int cus = 256;
  ce_gemm256_workspace(stream, &ws, &ws_bytes, &cus);
  return ce_gemm_bf16_tile_rows(M, N, K, cus, ws != nullptr ? (long long)ws_bytes : 0);
}

CE_API int ce_gemm_seg_bf16(const void* A, const void* W, void* C, const float* bias, int epilogue, const float* gate,
                                const void* res, int M, int N, int K, int lda, int ldw, int ldc, int ldres, int gate_rows,
                                int a_seg_k, long long a_seg_stride, int w_seg_k, long long w_seg_stride, hipStream_t stream) {
  if (!A || !W || !C) return CE_ERR_ARG;
  if (epilogue == EPI_BIAS_T) {
    // C^T is stored (C is [N][ldc]).  Mathematically the EPI_BIAS_ROW product with the operands swapped; run as written - M the token count -
    // on the 384- / 288-row kernel when that is what the dispatcher would give (M, N, K) anyway and its last round runs whole (the transposed
    // store has no split-K reduce), as the swapped EPI_BIAS_ROW product otherwise: the same sums either way (same products, same k order).
    if (!bias || M <= 0 || N <= 0 || K <= 0 || (K % BK)) return CE_ERR_ARG;
    if ((lda & 7) || (ldw & 7) || (ldc & 7)) return CE_ERR_ALIGN;
    const bool plain = (a_seg_k <= 0 || a_seg_k >= K) && (w_seg_k <= 0 || w_seg_k >= K);
    if (plain && (M & 7) == 0 && (N & 7) == 0 && (long long)M * N >= 256ll * 256 * 128 && (long long)N * ldc * 2 < (1ll << 32) &&
        ce_gemm256_supported(M, N, K, lda, ldw) && (g_gemm_variant == -1 || g_gemm_variant == 6 || g_gemm_variant == 7)) {
      float* ws = nullptr;
      size_t ws_bytes = 0;
      int cus = 256;
      ce_gemm256_workspace(stream, &ws, &ws_bytes, &cus);
      const int rows = g_gemm_variant == 6 ? 384 : g_gemm_variant == 7 ? 288 : ce_gemm_bf16_tile_rows(M, N, K, cus, ws != nullptr ? (long long)ws_bytes : 0);
      if (rows == 384 || rows == 288) {
        const long long nwg = (long long)((M + rows - 1) / rows) * ((N + 255) / 256);
        const int tail = (int)(nwg % cus);
        if (g_gemm_variant != -1 || tail == 0 || tail * 10 >= cus * 9)
          return (rows == 384 ? ce_gemm384_launch : ce_gemm288_launch)(A, W, C, bias, EPI_BIAS_T, nullptr, nullptr, M, N, K, lda, ldw, ldc, 0, 0, 0, 0, 0, 0,
                                                                      stream);
      }
    }
    if (!plain) return CE_ERR_SHAPE;
    return ce_gemm_seg_bf16(W, A, C, bias, EPI_BIAS_ROW, nullptr, nullptr, N, M, K, ldw, lda, ldc, 0, 0, 0, 0, 0, 0, stream);
  }
  if (w_seg_k < 0 || (w_seg_k > 0 && w_seg_k < K && ((w_seg_k % BK) || (K % w_seg_k) || (w_seg_stride & 7)))) return CE_ERR_SHAPE;
  if (w_seg_k >= K) w_seg_k = 0;
  if (a_seg_k < 0 || (a_seg_k > 0 && a_seg_k < K && ((a_seg_k % BK) || (K % a_seg_k) || (a_seg_stride & 7)))) return CE_ERR_SHAPE;
  if (a_seg_k >= K) a_seg_k = 0;
  if (M <= 0 || N <= 0 || K <= 0 || (K % BK) || (N & 7)) return CE_ERR_SHAPE;
  if ((lda & 7) || (ldw & 7) || (ldc & 7)) return CE_ERR_ALIGN;
  if ((epilogue == EPI_GATE_RES || epilogue == EPI_MUL) && (!res || (ldres & 7))) return CE_ERR_ARG;
  if (epilogue < 0 || epilogue > 6 || (epilogue == EPI_BIAS_ROW && !bias)) return CE_ERR_ARG;
  if (epilogue != EPI_F32 && epilogue != EPI_MUL) {
    const bool big = (long long)M * N >= 256ll * 256 * 128;  // enough 256x256 tiles to fill half the chip
    const bool want = g_gemm_variant >= 1 || (g_gemm_variant == -1 && big);
    if ((want || w_seg_k) && ce_gemm256_supported(M, N, K, lda, ldw)) {
      // the 256-tile kernel has two main loops: one wave per SIMD (ce_gemm256w4.hip; the default: +3...5 % on the step's shapes,
      // profiles/r03_gemm_variants_ab.txt) and the 8-wave / 8-phase loop of ce_gemm256.hip (variants 1, 2)
      const int rows_auto = g_gemm_variant == -1 ? auto_tile_rows(M, N, K, stream) : 0;
      if (g_gemm_variant == 6 || rows_auto == 384)  // the 384 x 256 macro tile (ce_gemm384.hip)
        return ce_gemm384_launch(A, W, C, bias, epilogue, gate, res, M, N, K, lda, ldw, ldc, ldres, gate_rows, a_seg_k, a_seg_stride,
                                 w_seg_k, w_seg_stride, stream);
      if (g_gemm_variant == 7 || rows_auto == 288)  // the 288 x 256 macro tile (the same kernel, 144 x 128 wave tiles)
        return ce_gemm288_launch(A, W, C, bias, epilogue, gate, res, M, N, K, lda, ldw, ldc, ldres, gate_rows, a_seg_k, a_seg_stride,
                                 w_seg_k, w_seg_stride, stream);
      if (g_gemm_variant == 1 || g_gemm_variant == 2)
        return ce_gemm256_launch(A, W, C, bias, epilogue, gate, res, M, N, K, lda, ldw, ldc, ldres, gate_rows, a_seg_k, a_seg_stride,
                                 w_seg_k, w_seg_stride, stream);
      return ce_gemm256w4_launch(A, W, C, bias, epilogue, gate, res, M, N, K, lda, ldw, ldc, ldres, gate_rows, a_seg_k, a_seg_stride,
                                 w_seg_k, w_seg_stride, g_gemm_variant == 3 ? 3 : g_gemm_variant == 5 ? 1 : 2, stream);
    }
  }
  if (w_seg_k) return CE_ERR_SHAPE;
  const int a_seg_tiles = a_seg_k > 0 ? a_seg_k / BK : 0;
  const long long a_seg_extra = a_seg_k > 0 ? a_seg_stride - a_seg_k : 0;
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  dim3 grid(tiles_m * tiles_n), block(256);
#define CE_LAUNCH(E)                                                                                              \
  hipLaunchKernelGGL(gemm_bf16_128<E>, grid, block, 0, stream, (const bf16*)A, (const bf16*)W, (bf16*)C, bias, gate, \
                     (const bf16*)res, M, N, K, lda, ldw, ldc, ldres, gate_rows, tiles_m, tiles_n, 1, 0ll, 0ll, 0ll, 0ll, 0ll, 0ll, \
                     a_seg_tiles, a_seg_extra)
  switch (epilogue) {
    case EPI_BIAS: CE_LAUNCH(EPI_BIAS); break;
    case EPI_BIAS_GELU: CE_LAUNCH(EPI_BIAS_GELU); break;
    case EPI_GATE_RES: CE_LAUNCH(EPI_GATE_RES); break;
    case EPI_BIAS_GELU_ERF: CE_LAUNCH(EPI_BIAS_GELU_ERF); break;
    case EPI_F32: CE_LAUNCH(EPI_F32); break;
    case EPI_MUL: CE_LAUNCH(EPI_MUL); break;
    case EPI_BIAS_ROW: CE_LAUNCH(EPI_BIAS_ROW); break;
    default: return CE_ERR_ARG;
  }
#undef CE_LAUNCH
  return (int)hipGetLastError();
}

CE_API int ce_gemm_aseg_bf16(const void* A, const void* W, void* C, const float* bias, int epilogue, const float* gate,
                                 const void* res, int M, int N, int K, int lda, int ldw, int ldc, int ldres, int gate_rows,
                                 int a_seg_k, long long a_seg_stride, hipStream_t stream) {
  return ce_gemm_seg_bf16(A, W, C, bias, epilogue, gate, res, M, N, K, lda, ldw, ldc, ldres, gate_rows, a_seg_k, a_seg_stride, 0, 0, stream);
}

CE_API int ce_gemm_bf16(const void* A, const void* W, void* C, const float* bias, int epilogue, const float* gate,
                            const void* res, int M, int N, int K, int lda, int ldw, int ldc, int ldres, int gate_rows,
                            hipStream_t stream) {
  return ce_gemm_seg_bf16(A, W, C, bias, epilogue, gate, res, M, N, K, lda, ldw, ldc, ldres, gate_rows, 0, 0, 0, 0, stream);
}

// batch0 x batch1 independent products with two-level element strides (e.g. head within sample): operand z = (z0, z1) is
// A + z0 sA0 + z1 sA1, W + z0 sW0 + z1 sW1, C + z0 sC0 + z1 sC1 (C strides in elements of C's type).  Epilogues EPI_BIAS
// and EPI_F32 only; always the 128-tile kernel (these are the per-head attention products of the encoders).
CE_API int ce_gemm_batched_bf16(const void* A, const void* W, void* C, const float* bias, int epilogue, int M, int N, int K,
                                    int lda, int ldw, int ldc, int batch0, int batch1, long long sA0, long long sA1, long long sW0,
                                    long long sW1, long long sC0, long long sC1, hipStream_t stream) {
  if (!A || !W || !C) return CE_ERR_ARG;
  if (M <= 0 || N <= 0 || K <= 0 || (K % BK) || (N & 7) || batch0 <= 0 || batch1 <= 0 || (long long)batch0 * batch1 > 65535)
    return CE_ERR_SHAPE;
  if ((lda & 7) || (ldw & 7) || (ldc & 7) || (sA0 & 7) || (sA1 & 7) || (sW0 & 7) || (sW1 & 7) || (sC0 & 7) || (sC1 & 7))
    return CE_ERR_ALIGN;
  if (epilogue != EPI_BIAS && epilogue != EPI_F32) return CE_ERR_ARG;
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  dim3 grid(tiles_m * tiles_n, batch0 * batch1), block(256);
#define CE_LAUNCH(E)                                                                                                 \
  hipLaunchKernelGGL(gemm_bf16_128<E>, grid, block, 0, stream, (const bf16*)A, (const bf16*)W, (bf16*)C, bias, nullptr, \
                     nullptr, M, N, K, lda, ldw, ldc, 0, 0, tiles_m, tiles_n, batch0, sA0, sA1, sW0, sW1, sC0, sC1, 0, 0ll)
  if (epilogue == EPI_BIAS) CE_LAUNCH(EPI_BIAS); else CE_LAUNCH(EPI_F32);
#undef CE_LAUNCH
  return (int)hipGetLastError();
}
