// Wan-2.1 VAE kernels (K20/K21): implicit-GEMM causal conv3d / conv2d on MFMA over channels-last frames,
// plus the HBM-bound companions (RMS_norm+SiLU, nearest 2x upsample, row softmax for the mid-block attention).
// Reference spec: chronoedit/_src/tokenizers/wan2pt1.py (CausalConv3d :42-60, RMS_norm :63-75, Resample :86-165,
// ResidualBlock :186-220, AttentionBlock :223-259); call sites pipeline_chronoedit.py:442,776-781.
//
// Data layout: every activation frame is channels-last with a one-pixel ZERO border, [H+2][W+2][C] bf16.  Spatial zero
// padding of the reference's convs is then just addressing; temporal causal padding is a list of frame pointers
// (cache frames of the previous chunk or a shared zero frame in front of the chunk's own frames), so nothing is ever
// concatenated or padded in memory.
//
// conv_igemm: out[t][h][w][co] = bias[co] + sum_{kt,kh,kw,ci} W[co][(kt,kh,kw)][ci] * in[t*st+kt][h*ss+kh+oh][w*ss+kw+ow][ci]
//   GEMM view M = T_out*H_out*W_out output pixels, N = Cout, K = taps * Cin, both operands K(=ci)-contiguous.
//   128 x BN x 32 block tile, 4 waves x (32 rows x BN cols) of v_mfma_f32_16x16x32_bf16, register-staged double-buffered
//   LDS tiles (16-B chunk XOR swizzle), epilogue through LDS for 16-B stores.  Bound: bf16 MFMA for >= 192 channels,
//   HBM for the 96-channel full-resolution layers.
#include "ce_common.h"

namespace {

constexpr int CBM = 128, CBK = 32;
constexpr int MAX_FRAMES = 16;

struct ConvParams {
  const bf16* in_frames[MAX_FRAMES];
  bf16* out_frames[MAX_FRAMES];
  const bf16* res_frames[MAX_FRAMES];
  const bf16* weight;   // [Cout][taps][Cin]
  const float* bias;
  int n_out_frames, Cin, Cout, KT, KH, KW, st, ss;
  int H_out, W_out, in_Wp, in_off_h, in_off_w;
  int out_Wp, out_border, out_cstride, out_coff;
  int has_res;
};

__device__ __forceinline__ int cswz(int row, int chunk) { return chunk ^ ((row >> 2) & 3); }
typedef __attribute__((address_space(3))) void lds_void_c;  // LDS-DMA destinations

template <int BN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvParams p) {
  constexpr int NF = BN / 16;                 // n fragments per wave
  constexpr int A_BYTES = CBM * CBK * 2;      // 8 KiB
  constexpr int W_BYTES = BN * CBK * 2;
  constexpr int CROW = BN * 2 + 16;           // padded epilogue row
  constexpr int TILE_BYTES = 2 * (A_BYTES + W_BYTES);
  constexpr int SMEM = TILE_BYTES > CBM * CROW ? TILE_BYTES : CBM * CROW;
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
  __shared__ const bf16* s_in[MAX_FRAMES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  if (tid < MAX_FRAMES) s_in[tid] = p.in_frames[tid];
  __syncthreads();

  const int HW = p.H_out * p.W_out;
  const int M = p.n_out_frames * HW;
  const int tiles_n = (p.Cout + BN - 1) / BN;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  const int m0 = tm * CBM, n0 = tn * BN;
  const int taps = p.KT * p.KH * p.KW;
  const int cchunks = p.Cin / CBK;
  const int KTILES = taps * cchunks;

  // A staging: 128 rows x 4 chunks(16 B) = 512 chunks, 2 per thread: rows (tid>>2) and (tid>>2)+64, chunk tid&3
  const int a_ck = tid & 3;
  int a_t[2], a_sp[2];   // output frame index, spatial element offset (without tap) into the bordered input frame
  int a_lds[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (tid >> 2) + 64 * i;
    const int m = min(m0 + r, M - 1);
    const int t = m / HW, rem = m - t * HW;
    const int h = rem / p.W_out, w = rem - h * p.W_out;
    a_t[i] = t * p.st;
    a_sp[i] = ((h * p.ss + p.in_off_h) * p.in_Wp + (w * p.ss + p.in_off_w)) * p.Cin + a_ck * 8;
    a_lds[i] = r * (CBK * 2) + (cswz(r, a_ck) << 4);
  }
  // W staging: BN rows x 4 chunks; BN*4/256 per thread
  constexpr int WREP = (BN * 4 + 255) / 256;
  const bf16* w_src[WREP];
  int w_lds[WREP];
  bool w_on[WREP];
#pragma unroll
  for (int i = 0; i < WREP; ++i) {
    const int c = tid + 256 * i;
    const int r = c >> 2, ck = c & 3;
    w_on[i] = r < BN;
    const int n = min(n0 + r, p.Cout - 1);
    w_src[i] = p.weight + (size_t)n * taps * p.Cin + ck * 8;
    w_lds[i] = r * (CBK * 2) + (cswz(r, ck) << 4);
  }

  f32x4 acc[2][NF];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 ra[2], rw[WREP];
  auto load_tile = [&](int kt) {
    const int tap = kt / cchunks, cc = kt - tap * cchunks;
    const int kw = tap % p.KW, kh = (tap / p.KW) % p.KH, kti = tap / (p.KW * p.KH);
    const int tap_off = (kh * p.in_Wp + kw) * p.Cin + cc * CBK;
#pragma unroll
    for (int i = 0; i < 2; ++i) ra[i] = *reinterpret_cast<const u32x4*>(s_in[a_t[i] + kti] + a_sp[i] + tap_off);
#pragma unroll
    for (int i = 0; i < WREP; ++i)
      if (w_on[i]) rw[i] = *reinterpret_cast<const u32x4*>(w_src[i] + (size_t)tap * p.Cin + cc * CBK);
  };
  auto store_tile = [&](int buf) {
    unsigned char* sA = smem + buf * (A_BYTES + W_BYTES);
    unsigned char* sW = sA + A_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<u32x4*>(sA + a_lds[i]) = ra[i];
#pragma unroll
    for (int i = 0; i < WREP; ++i)
      if (w_on[i]) *reinterpret_cast<u32x4*>(sW + w_lds[i]) = rw[i];
  };

  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int kt = 0; kt < KTILES; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < KTILES) load_tile(kt + 1);
    const unsigned char* sA = smem + cur * (A_BYTES + W_BYTES);
    const unsigned char* sW = sA + A_BYTES;
    bf16x8 af[2], wf[NF];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = wave * 32 + i * 16 + fr;
      af[i] = *reinterpret_cast<const bf16x8*>(sA + r * (CBK * 2) + (cswz(r, fg) << 4));
    }
#pragma unroll
    for (int j = 0; j < NF; ++j) {
      const int r = j * 16 + fr;
      wf[j] = *reinterpret_cast<const bf16x8*>(sW + r * (CBK * 2) + (cswz(r, fg) << 4));
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NF; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], wf[j], acc[i][j], 0, 0, 0);
    if (kt + 1 < KTILES) store_tile(cur ^ 1);
    __syncthreads();
  }

  // epilogue: bf16(acc + bias) -> LDS rows -> 16-B chunks (+ residual) -> bordered output frame
#pragma unroll
  for (int j = 0; j < NF; ++j) {
    const int cl = j * 16 + fr;
    const int n = n0 + cl;
    const float bv = (p.bias != nullptr && n < p.Cout) ? p.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rl = wave * 32 + i * 16 + fg * 4 + r;
        *reinterpret_cast<bf16*>(smem + rl * CROW + cl * 2) = (bf16)(acc[i][j][r] + bv);
      }
  }
  __syncthreads();
  constexpr int CPR = BN / 8;  // 16-B chunks per row
  for (int c = tid; c < CBM * CPR; c += 256) {
    const int rl = c / CPR, cc = c % CPR;
    const int m = m0 + rl, n = n0 + cc * 8;
    if (m < M && n < p.Cout) {
      const int t = m / HW, rem = m - t * HW;
      const int h = rem / p.W_out, w = rem - h * p.W_out;
      const size_t off = ((size_t)(h + p.out_border) * p.out_Wp + (w + p.out_border)) * p.out_cstride + p.out_coff + n;
      u32x4 v = *reinterpret_cast<const u32x4*>(smem + rl * CROW + cc * 16);
      if (p.has_res) {
        const u32x4 rv = *reinterpret_cast<const u32x4*>(p.res_frames[t] + off);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = pack_bf16(bf16lo(rv[q]) + bf16lo(v[q]), bf16hi(rv[q]) + bf16hi(v[q]));
      }
      *reinterpret_cast<u32x4*>(p.out_frames[t] + off) = v;
    }
  }
}

// ---- RMS_norm over channels (+ SiLU), channels-last, in -> out (both bordered frames or plain rows) ---------------
// y = silu( x / max(||x||_2, 1e-12) * sqrt(C) * gamma )     (F.normalize semantics, wan2pt1.py:74)
// HBM-bound (one read, one write of the activation): 16 lanes per pixel, LPA of them active, each holding PER 16-byte chunks (8
// channels) of the pixel in registers - chunk i * LPA + sub - so a pixel is read ONCE, in 16-byte accesses; a workgroup takes 16 PPG
// consecutive pixels of ONE image row (blockIdx.x = frame * H + row: no per-pixel division), every lane with PPG loads in flight.
// (The first version read every pixel twice in 4-byte accesses behind two 64-bit divisions: 2.0 TB/s on the full-resolution frames,
// 15 % of a 720p decode.)
template <int PER, int PPG>
__global__ __launch_bounds__(256) void rms_silu_kernel(const bf16* __restrict__ x, bf16* __restrict__ y,
                                                        const float* __restrict__ gamma, int C, int LPA, int H, int W, int in_Wp,
                                                        int in_border, int out_Wp, int out_border, int apply_silu) {
  const int sub = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int t = blockIdx.x / H, h = blockIdx.x - t * H;
  const size_t in_row = ((size_t)t * (H + 2 * in_border) + h + in_border) * in_Wp + in_border;
  const size_t out_row = ((size_t)t * (H + 2 * out_border) + h + out_border) * out_Wp + out_border;
  const int w0 = blockIdx.y * (16 * PPG) + grp;
  u32x4 v[PPG][PER];
  float ss[PPG];
#pragma unroll
  for (int j = 0; j < PPG; ++j) {
    const int w = w0 + 16 * j;
    const bool on = w < W && sub < LPA;
    const bf16* xp = x + (in_row + min(w, W - 1)) * C;
#pragma unroll
    for (int i = 0; i < PER; ++i) v[j][i] = on ? *reinterpret_cast<const u32x4*>(xp + (i * LPA + sub) * 8) : u32x4{0u, 0u, 0u, 0u};
  }
#pragma unroll
  for (int j = 0; j < PPG; ++j) {
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) a += bf16lo(v[j][i][q]) * bf16lo(v[j][i][q]) + bf16hi(v[j][i][q]) * bf16hi(v[j][i][q]);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) a += __shfl_xor(a, o, 16);
    ss[j] = a;
  }
  if (sub >= LPA) return;
  f32x4 g0[PER], g1[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int c = (i * LPA + sub) * 8;
    g0[i] = *reinterpret_cast<const f32x4*>(gamma + c);
    g1[i] = *reinterpret_cast<const f32x4*>(gamma + c + 4);
  }
#pragma unroll
  for (int j = 0; j < PPG; ++j) {
    const int w = w0 + 16 * j;
    if (w >= W) break;
    const float scale = sqrtf((float)C) / fmaxf(sqrtf(ss[j]), 1e-12f);
    bf16* yp = y + (out_row + w) * C;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      u32x4 o;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float a = bf16lo(v[j][i][q]) * scale * (q < 2 ? g0[i][2 * q] : g1[i][2 * q - 4]);
        float b = bf16hi(v[j][i][q]) * scale * (q < 2 ? g0[i][2 * q + 1] : g1[i][2 * q - 3]);
        if (apply_silu) {
          a = silu_fast(a);
          b = silu_fast(b);
        }
        o[q] = pack_bf16(a, b);
      }
      *reinterpret_cast<u32x4*>(yp + (i * LPA + sub) * 8) = o;
    }
  }
}

// ---- nearest-exact 2x spatial upsample, channels-last bordered frames (wan2pt1.py:78-83) ----------------------------
__global__ void upsample2x_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, long long nchunks, int C8, int H, int W,
                                  int in_Wp, int out_Wp) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nchunks) return;
  const int c = idx % C8;
  long long r = idx / C8;
  const int wo = r % (2 * W);
  r /= 2 * W;
  const int ho = r % (2 * H);
  const long long t = r / (2 * H);
  const u32x4 v = *reinterpret_cast<const u32x4*>(x + (((t * (H + 2) + (ho >> 1) + 1) * in_Wp + (wo >> 1) + 1) * (long long)C8 + c) * 8);
  *reinterpret_cast<u32x4*>(y + (((t * (2 * H + 2) + ho + 1) * out_Wp + wo + 1) * (long long)C8 + c) * 8) = v;
}

// ---- row softmax: fp32 scores [M][ld] -> bf16 probabilities [M][ldp] (columns >= n zero-filled up to npad) ----------
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s, bf16* __restrict__ pr, int n, int npad, int ld,
                                                           int ldp, float scale) {
  __shared__ float red[8];
  const float* row = s + (size_t)blockIdx.x * ld;
  bf16* out = pr + (size_t)blockIdx.x * ldp;
  const int tid = threadIdx.x;
  float mx = -3.0e38f;
  for (int i = tid; i < n; i += 256) mx = fmaxf(mx, row[i]);
  mx = wave_max(mx);
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
  for (int i = tid; i < n; i += 256) sum += __expf((row[i] - mx) * scale);
  sum = wave_sum(sum);
  if ((tid & 63) == 0) red[4 + (tid >> 6)] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
  for (int i = tid; i < npad; i += 256) out[i] = (bf16)(i < n ? __expf((row[i] - mx) * scale) * inv : 0.f);
}

}  // namespace

CE_API int ce_conv_igemm_bf16(const void* const* in_frames, int n_in_frames, const void* weight, const float* bias,
                                  void* const* out_frames, int n_out_frames, const void* const* res_frames, int Cin, int Cout,
                                  int KT, int KH, int KW, int st, int ss, int H_out, int W_out, int in_Wp, int in_off_h,
                                  int in_off_w, int out_Wp, int out_border, int out_cstride, int out_coff, hipStream_t stream) {
  if (!in_frames || !weight || !out_frames) return CE_ERR_ARG;
  if (n_in_frames <= 0 || n_in_frames > MAX_FRAMES || n_out_frames <= 0 || n_out_frames > MAX_FRAMES) return CE_ERR_SHAPE;
  if ((Cin % CBK) || (Cout & 7) || (out_cstride & 7) || (out_coff & 7) || KT < 1 || KH < 1 || KW < 1) return CE_ERR_SHAPE;
  if ((n_out_frames - 1) * st + KT > n_in_frames) return CE_ERR_SHAPE;
  ConvParams p;
  for (int i = 0; i < MAX_FRAMES; ++i) {
    p.in_frames[i] = (const bf16*)in_frames[i < n_in_frames ? i : n_in_frames - 1];
    p.out_frames[i] = (bf16*)out_frames[i < n_out_frames ? i : n_out_frames - 1];
    p.res_frames[i] = res_frames ? (const bf16*)res_frames[i < n_out_frames ? i : n_out_frames - 1] : nullptr;
  }
  p.weight = (const bf16*)weight;
  p.bias = bias;
  p.n_out_frames = n_out_frames;
  p.Cin = Cin; p.Cout = Cout; p.KT = KT; p.KH = KH; p.KW = KW; p.st = st; p.ss = ss;
  p.H_out = H_out; p.W_out = W_out; p.in_Wp = in_Wp; p.in_off_h = in_off_h; p.in_off_w = in_off_w;
  p.out_Wp = out_Wp; p.out_border = out_border; p.out_cstride = out_cstride; p.out_coff = out_coff;
  p.has_res = res_frames != nullptr;
  const long long M = (long long)n_out_frames * H_out * W_out;
  const int tiles_m = (int)((M + CBM - 1) / CBM);
  if (Cout <= 32) {
    hipLaunchKernelGGL(conv_igemm_kernel<32>, dim3(tiles_m * ((Cout + 31) / 32)), dim3(256), 0, stream, p);
  } else {
    hipLaunchKernelGGL(conv_igemm_kernel<128>, dim3(tiles_m * ((Cout + 127) / 128)), dim3(256), 0, stream, p);
  }
  return (int)hipGetLastError();
}

// ---- the decoder's head conv: 3 x 3 x 3, stride 1, 96 -> 3 channels at full resolution (wan2pt1.py:401-403, Decoder3d.head) -------
// As an implicit GEMM it has N = 3 (padded to 8, computed as a 32-wide tile): 2.7 ms at 720p for ~1 GB of input.  Here the three
// kernel ROWS ride in the matrix instruction's idle output rows instead: A = weights with row n = (kh, co) = 3 x 4 of the 16,
// B = 16 consecutive pixels of ONE input row rho, K = 32 input channels.  P[rho][(kh, co)][pixel] summed over (kt, kw, channel
// chunk) is then every contribution input row rho makes to the three output rows rho - kh at once, and out[h] = P[h][kh = 0] +
// P[h + 1][kh = 1] + P[h + 2][kh = 2] is three lanes' registers (row n of D lives in lane group n >> 2): two cross-lane moves per
// value at the very end.  A wave owns 8 output rows x 16 columns = 10 input rows: 27 x 10 MFMAs (a third of the per-tap count), the
// B fragments straight from global memory - 16-byte loads, a pixel row's 18 x 192 B strip is touched by nine loads in one burst and
// comes out of the L1 - with the 27 weight fragments of the workgroup in LDS (27 KiB) and the nine of the current kt in registers.
// Workgroup: 4 waves = 8 rows x 64 columns (162 registers: three workgroups per CU).
namespace {
constexpr int HEAD_ROWS = 8, HEAD_COLS = 64;

__global__ __launch_bounds__(256) void conv_head_kernel(ConvParams p, int H_tiles, int W_tiles) {
  __shared__ __attribute__((aligned(16))) unsigned char wsm[27 * 1024];
  __shared__ const bf16* s_in[MAX_FRAMES];
  __shared__ bf16* s_out[MAX_FRAMES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  constexpr int CIN = 96, CCH = 3;
  const int taps = p.KT * 9;
  if (tid < MAX_FRAMES) {
    s_in[tid] = p.in_frames[tid];
    s_out[tid] = p.out_frames[tid];
  }
  // weight fragments, lane-linear: fragment (kt, kw, c), lane (n = (kh, co), g) <- W[co][(kt, kh, kw)][32 c + 8 g ...]
  for (int i = tid; i < p.KT * 9 * 64; i += 256) {
    const int f = i >> 6, l = i & 63, n = l & 15, g = l >> 4;
    const int kt = f / 9, kw = (f % 9) / CCH, c = f % CCH, kh = n >> 2, co = n & 3;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (kh < 3 && co < p.Cout) v = *reinterpret_cast<const u32x4*>(p.weight + ((size_t)co * taps + (kt * 3 + kh) * 3 + kw) * CIN + c * 32 + g * 8);
    *reinterpret_cast<u32x4*>(wsm + (size_t)i * 16) = v;
  }
  __syncthreads();

  const int wb = blockIdx.x % W_tiles, hb = (blockIdx.x / W_tiles) % H_tiles, t = blockIdx.x / (W_tiles * H_tiles);
  const int h0 = hb * HEAD_ROWS, w0 = wb * HEAD_COLS + wave * 16;
  if (w0 >= p.W_out) return;
  const int Hp = p.H_out + 2;
  const int wcol = min(w0 + fr, p.W_out - 1);  // (clamped columns are computed and not stored)
  int roff[HEAD_ROWS + 2];
#pragma unroll
  for (int r = 0; r < HEAD_ROWS + 2; ++r) roff[r] = (min(h0 + r, Hp - 1) * p.in_Wp + wcol) * CIN + fg * 8;

  f32x4 acc[HEAD_ROWS + 2];
#pragma unroll
  for (int r = 0; r < HEAD_ROWS + 2; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};

  const uint32_t frame_bytes = (uint32_t)Hp * (uint32_t)p.in_Wp * CIN * 2u;
  for (int kt = 0; kt < p.KT; ++kt) {
    // the frame as a buffer resource (uniform base, 32-bit byte offsets): buffer loads count on vmcnt alone - a pointer read back from
    // LDS is a generic one to hipcc, and flat loads made it wait for every single fragment
    const uint64_t fp = reinterpret_cast<uint64_t>(s_in[t + kt]);
    const uint64_t fpu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(fp >> 32)) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)fp);
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(fpu), 0, frame_bytes, 0x00020000);
    bf16x8 wf[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) wf[q] = *reinterpret_cast<const bf16x8*>(wsm + ((size_t)(kt * 9 + q) * 64 + lane) * 16);
    u32x4 bq[2][9];
#pragma unroll
    for (int q = 0; q < 9; ++q) bq[0][q] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, roff[0] * 2, ((q / CCH) * CIN + (q % CCH) * 32) * 2, 0);
#pragma unroll
    for (int r = 0; r < HEAD_ROWS + 2; ++r) {
      if (r + 1 < HEAD_ROWS + 2) {
#pragma unroll
        for (int q = 0; q < 9; ++q)
          bq[(r + 1) & 1][q] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, roff[r + 1] * 2, ((q / CCH) * CIN + (q % CCH) * 32) * 2, 0);
      }
      __builtin_amdgcn_sched_barrier(0);  // row r+1's nine loads are in flight BEFORE row r's MFMAs (hipcc otherwise walks q outermost)
#pragma unroll
      for (int q = 0; q < 9; ++q)
        acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[q], __builtin_bit_cast(bf16x8, bq[r & 1][q]), acc[r], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // out[h0 + j] = P[j][kh 0] + P[j + 1][kh 1] + P[j + 2][kh 2]: lane group fg holds kernel row kh = fg of every P
  f32x4 bias = {0.f, 0.f, 0.f, 0.f};
  if (p.bias != nullptr) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i < p.Cout) bias[i] = p.bias[i];
  }
  bf16* __restrict__ out = s_out[t];
#pragma unroll
  for (int j = 0; j < HEAD_ROWS; ++j) {
    f32x4 v = acc[j];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] += __shfl_down(acc[j + 1][i], 16) + __shfl_down(acc[j + 2][i], 32) + bias[i];
    const int h = h0 + j, w = w0 + fr;
    if (fg == 0 && h < p.H_out && w < p.W_out) {
      const size_t o = ((size_t)(h + p.out_border) * p.out_Wp + (w + p.out_border)) * p.out_cstride + p.out_coff;
      if (p.out_cstride - p.out_coff >= 8)
        *reinterpret_cast<u32x4*>(out + o) = u32x4{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), 0u, 0u};
      else
        *reinterpret_cast<u32x2*>(out + o) = u32x2{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
    }
  }
}
}  // namespace

// 3 x 3 x 3 (KT = 3) or 1 x 3 x 3 (KT = 1) stride-1 conv of 96 input channels onto Cout <= 4 channels (weight [>= Cout][KT*9][96]).
// Frames, addressing and result as ce_conv_igemm_bf16 with st = ss = 1, in_off = 0 and no residual; channels Cout .. 3 (and .. 7 when the
// pixel stride leaves room for them) are written as zeros.
CE_API int ce_conv3d_head_bf16(const void* const* in_frames, int n_in_frames, const void* weight, const float* bias,
                                   void* const* out_frames, int n_out_frames, int Cin, int Cout, int KT, int H_out, int W_out, int in_Wp,
                                   int out_Wp, int out_border, int out_cstride, int out_coff, hipStream_t stream) {
  if (!in_frames || !weight || !out_frames) return CE_ERR_ARG;
  if (n_in_frames <= 0 || n_in_frames > MAX_FRAMES || n_out_frames <= 0 || n_out_frames > MAX_FRAMES) return CE_ERR_SHAPE;
  if (Cin != 96 || Cout < 1 || Cout > 4 || (KT != 1 && KT != 3) || (out_cstride & 3) || (out_coff & 3) || out_cstride - out_coff < 4 ||
      n_out_frames - 1 + KT > n_in_frames || in_Wp < W_out + 2 || H_out < 1 || W_out < 1)
    return CE_ERR_SHAPE;
  ConvParams p;
  for (int i = 0; i < MAX_FRAMES; ++i) {
    p.in_frames[i] = (const bf16*)in_frames[i < n_in_frames ? i : n_in_frames - 1];
    p.out_frames[i] = (bf16*)out_frames[i < n_out_frames ? i : n_out_frames - 1];
    p.res_frames[i] = nullptr;
  }
  p.weight = (const bf16*)weight;
  p.bias = bias;
  p.n_out_frames = n_out_frames;
  p.Cin = Cin; p.Cout = Cout; p.KT = KT; p.KH = 3; p.KW = 3; p.st = 1; p.ss = 1;
  p.H_out = H_out; p.W_out = W_out; p.in_Wp = in_Wp; p.in_off_h = 0; p.in_off_w = 0;
  p.out_Wp = out_Wp; p.out_border = out_border; p.out_cstride = out_cstride; p.out_coff = out_coff;
  p.has_res = 0;
  const int H_tiles = (H_out + HEAD_ROWS - 1) / HEAD_ROWS, W_tiles = (W_out + HEAD_COLS - 1) / HEAD_COLS;
  hipLaunchKernelGGL(conv_head_kernel, dim3(n_out_frames * H_tiles * W_tiles), dim3(256), 0, stream, p, H_tiles, W_tiles);
  return (int)hipGetLastError();
}

// ---- 3 x 3 (x 3) stride-1 convolutions of >= 128 output channels on the 256 x 256 x 64 LDS-DMA GEMM (ce_gemm256w4.hip) -------------
// On bordered channels-last frames a stride-1 3 x 3 x 3 convolution IS a GEMM whose A rows are linear in memory: count output
// positions on the PADDED grid, p = (t Hp + hp) Wp + wp, and tap (kt, kh, kw) of output position p reads input position
// p + kt Hp Wp + (kh - 1) Wp + (kw - 1) of a stack of frames - a constant offset.  So A = the input stack itself with lda = Cin, row
// r <-> position r + Wp + 1 (the first interior pixel; every address is then >= the stack's base), K = 27 Cin walked tap by tap: kw runs
// on contiguously (the next pixel), kh jumps a pixel row, kt a frame - a K-segmented operand on two nested levels, one scalar offset
// per K-tile.  C = the output stack, linear in the same row index.  Rows that fall on border positions (under 1 % of them) are computed
// and then zeroed again by zero_border_kernel: the border is the next layer's padding.  K-tiles are consumed in pairs: an odd count is
// padded by one tile whose weight columns are zero and whose A tile lies three frames (KT = 3) or three pixel rows plus one frame
// (KT = 1) further on - the caller keeps one zeroed slack frame behind the last input frame.  Cin = 96 (three pixels = 288 channels = 4.5
// K-tiles): every (kt, kh) run is rounded up to 5 tiles whose last 32 columns carry zero weights.
// Reference: CausalConv3d, wan2pt1.py:42-60 (the two front frames of the stack are its causal padding / feat_cache).
namespace {
__global__ __launch_bounds__(256) void zero_border_kernel(bf16* __restrict__ y, int T, int Hp, int Wp, int C8, int ld8) {
  // border pixels of a frame: rows 0 and Hp-1 (Wp each), columns 0 and Wp-1 of the rows between (2 (Hp - 2))
  const int per = 2 * Wp + 2 * (Hp - 2);
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)T * per * C8) return;
  const int c = (int)(idx % C8);
  const long long r = idx / C8;
  const int t = (int)(r / per), b = (int)(r % per);
  int hp, wp;
  if (b < Wp) { hp = 0; wp = b; }
  else if (b < 2 * Wp) { hp = Hp - 1; wp = b - Wp; }
  else { const int q = b - 2 * Wp; hp = 1 + (q >> 1); wp = (q & 1) ? Wp - 1 : 0; }
  *reinterpret_cast<u32x4*>(y + ((((size_t)t * Hp + hp) * Wp + wp) * ld8 + c) * 8) = u32x4{0u, 0u, 0u, 0u};
}
}  // namespace

// ---- 3 x 3 (x 3) stride-1 convolutions onto 96 output channels: the full-resolution layers of the VAE (round 6) ----------------------
// The same linear view as above (GEMM row r <-> bordered position r + Wp + 1, tap (kt, kh, kw) = input position r + kt Hp Wp + kh Wp + kw),
// but the A operand is NOT staged as GEMM tiles.  On the 256 x 96 macro tile of the large-tile GEMM these layers ran at 0.9 PFLOP/s because
// of bytes, not flops: a workgroup pulled 44 KB through its LDS per 3.1 MFLOP (70 flop / byte, ~21 B / cycle / CU at that rate - the
// L2 -> LDS ceiling), and two thirds of the A bytes were the SAME input pixels fetched once per kw: row r of the (kt, kh) run is the 3 Cin
// channels starting at position r, row r + 1 the same run one pixel on.  Here the LDS holds the input SLAB itself:
//   * a workgroup owns 512 consecutive positions x 96 output channels (four waves x (8 position blocks x 6 channel blocks) = 192 accumulators
//     per lane, one wave per SIMD); the K walk is (kt, kh, 32-channel chunk c): 3 KT Cin / 32 sub-stages of A image [528 positions][32 ch] (33 KB; 514
//     are read) + W image [3 kw][96 cout][32 ch] (18 KB), and the three kw taps read the A image at position offsets 0 / 1 / 2:
//     51 KB per 9.4 MFLOP = 181 flop / byte;
//   * both images arrive by LDS-DMA into a ring of three slots, two sub-stages ahead, one barrier per sub-stage (144 MFMAs per wave); rows
//     are 64 B, and position p of row r holds 16-byte chunk p ^ (2 (r >> 2 & 1)): a ds_read_b128 of 16 consecutive rows is conflict-free in
//     the instruction's four lane groups at ANY row offset (searched exhaustively), so the kw-shifted reads cost nothing;
//   * weights are read straight from the GEMM layout of ce_conv3d_gemm_bf16 (column ((kt 3 + kh) S + kw Cin + ci)): same entry point, same
//     operands, same result up to the summation order.
// Epilogue: bias, bf16 rounding, bf16(res + .), 8-byte stores (a lane owns 4 consecutive channels of a position); border positions are
// zeroed afterwards by zero_border_kernel, as for the GEMM route.
namespace {

constexpr int CR_TM = 510;  // positions per workgroup: with the two positions the kw taps reach past them the A image is 512 rows = 32 pieces
constexpr int CR_A_PIECES = 32, CR_A_BYTES = CR_A_PIECES * 1024;
constexpr int CR_B_PIECES = 18, CR_B_BYTES = CR_B_PIECES * 1024;
constexpr int CR_SLOT = CR_A_BYTES + CR_B_BYTES;  // 51 200
constexpr int CR_SMEM = 3 * CR_SLOT;              // 153 600 of 163 840

// NC = Cin / 32; NW waves: 4 (one per SIMD, 128 positions x 96 channels = 192 accumulators each) or 8 (two per SIMD, 64 positions = 96
// accumulators each: while one wave of a SIMD sits in the issue of an LDS-DMA piece - ~60-100 cycles each, 13 per sub-stage and wave with
// four waves - the other one feeds the matrix pipe)
template <int NC, int NW>
__device__ __forceinline__ void conv3x3_c96_body(const bf16* __restrict__ in, long long in_bytes, const bf16* __restrict__ wgt, int ldw,
                                                 const float* __restrict__ bias, bf16* __restrict__ out, const bf16* __restrict__ res, int rows, int KT,
                                                 int Wp, int FS, int S, int ocs, int T_out, int tiles_per_frame, const float* __restrict__ gamma,
                                                 int apply_silu, bf16* __restrict__ out2) {
  constexpr int Cin = 32 * NC;
  constexpr int MF = 32 / NW;                           // 16-position blocks per wave (8 or 4)
  constexpr int NA = CR_A_PIECES / NW;                  // A pieces per wave and sub-stage (8 or 4)
  constexpr int NB = (CR_B_PIECES + NW - 1) / NW;       // W issue slots per wave (5 or 3; the surplus ones repeat the last piece)
  constexpr int NSTEP = 3 * MF;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  // Tile order.  Workgroup b runs on XCD b % 8 (observed dispatch order; a speed matter only) and the eight L2s share nothing, while the
  // input rows a tile reads are read again by the tiles one image row up and down (kh) and by the same tile of the neighbouring output
  // frames (kt): dealt round-robin, tiles 8 apart (3.2 image rows at 720p) meet in no L2.  So an XCD takes a CONTIGUOUS run of the virtual
  // tile list, and the list walks the frames innermost: (position chunk, output frame) - the 32 CUs of an XCD then work on neighbouring
  // rows of all output frames at once (+ 9 % on the 4 x 720 x 1280 layer).
  const int per_xcd = gridDim.x >> 3;  // (the grid is a multiple of 8)
  const int v = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (v >= tiles_per_frame * T_out) return;
  const int t_out = v % T_out, chunk = v / T_out;
  const long long r0 = (long long)t_out * FS + (long long)chunk * CR_TM;
  const long long r_end = min((long long)rows, (long long)(t_out + 1) * FS);  // (the next frame's positions belong to its own tiles)
  const int nss = KT * 3 * NC;

  // LDS-DMA sources.  A: the input stack from this workgroup's first position on (positions past the stack read as zero: the range check
  // covers the VGPR offset, which carries the whole address here); W: always inside the matrix.
  const long long a_left = in_bytes - r0 * (Cin * 2);
  const auto a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(in + r0 * Cin), 0, (uint32_t)(a_left > 0xffffffffll ? 0xffffffffll : a_left), 0x00020000);
  const auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wgt, 0, (uint32_t)(96 * ldw * 2), 0x00020000);
  const int sw = 2 * ((lane >> 4) & 1);  // rows 16 P + (lane >> 2): (row >> 2) & 1 = (lane >> 4) & 1
  const int a_lane = (lane >> 2) * (Cin * 2) + (((lane & 3) ^ sw) << 4);
  int b_lane[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int P = min(wave + NW * j, CR_B_PIECES - 1);
    const int rho = 16 * P + (lane >> 2), kw = rho / 96, co = rho - 96 * kw;
    b_lane[j] = co * (ldw * 2) + kw * (Cin * 2) + (((lane & 3) ^ sw) << 4);
  }
  auto dma = [&](int i, int slot) __attribute__((always_inline)) {  // sub-stage i = ((kt 3 + kh) NC + c)
    const int c = i % NC, g = i / NC, kh = g % 3, kt = g / 3;
    const int a_off = (kt * FS + kh * Wp) * (Cin * 2) + c * 64, b_off = (g * S + 32 * c) * 2;
    unsigned char* const sb = smem + slot * CR_SLOT;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int P = wave + NW * j;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_void_c*)(sb + P * 1024), 16, a_lane + (P * 16 * (Cin * 2) + a_off), 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int P = min(wave + NW * j, CR_B_PIECES - 1);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void_c*)(sb + CR_A_BYTES + P * 1024), 16, b_lane[j], b_off, 0, 0);
    }
  };
  dma(0, 0);
  __builtin_amdgcn_sched_barrier(0);
  dma(min(1, nss - 1), 1);
  __builtin_amdgcn_sched_barrier(0);

  // fragment read offsets inside a slot: A (kw, mf) at a_rd[kw] + mf 1024; W (kw, n) at b_rd + (kw 96 + 16 n) 64
  int a_rd[3];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) a_rd[kw] = (16 * MF * wave + fr + kw) * 64 + ((fg ^ (2 * (((fr + kw) >> 2) & 1))) << 4);
  const int b_rd = CR_A_BYTES + fr * 64 + ((fg ^ (2 * ((fr >> 2) & 1))) << 4);

  f32x4 acc[MF][6];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf)
#pragma unroll
    for (int n = 0; n < 6; ++n) {
      acc[mf][n] = f32x4{0.f, 0.f, 0.f, 0.f};
      asm volatile("" : "+a"(acc[mf][n]));
    }

  int slot = 0;
  for (int i = 0; i < nss; ++i) {
    // this wave's pieces of sub-stage i have landed (those of i + 1 may fly); behind the barrier everybody's have, and everybody is done with
    // sub-stage i - 1, whose slot is refilled
    if constexpr (NA + NB == 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    static_assert(NA + NB == 13 || NA + NB == 7, "vmcnt immediates above");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    dma(min(i + 2, nss - 1), slot == 0 ? 2 : slot - 1);
    __builtin_amdgcn_sched_barrier(0);
    const unsigned char* const sb = smem + slot * CR_SLOT;
    // NSTEP steps (kw, mf) of six MFMAs; hipcc moves no load across a volatile asm, so the reads are pipelined by hand: the A fragment runs
    // three steps ahead, the six W fragments of the next kw are read during the current kw's steps
    auto read_a = [&](int s_) __attribute__((always_inline)) { return *reinterpret_cast<const bf16x8*>(sb + a_rd[s_ / MF] + (s_ % MF) * 1024); };
    auto read_w = [&](int kw, int n) __attribute__((always_inline)) { return *reinterpret_cast<const bf16x8*>(sb + b_rd + (kw * 96 + 16 * n) * 64); };
    bf16x8 wf[2][6], af[3];
#pragma unroll
    for (int n = 0; n < 6; ++n) wf[0][n] = read_w(0, n);
#pragma unroll
    for (int s_ = 0; s_ < 3; ++s_) af[s_] = read_a(s_);
#pragma unroll
    for (int s_ = 0; s_ < NSTEP; ++s_) {
      const int kw = s_ / MF, mf = s_ % MF;
      const bf16x8 a = af[s_ % 3];
#pragma unroll
      for (int n = 0; n < 6; ++n) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[mf][n]) : "v"(wf[kw & 1][n]), "v"(a));
      if (s_ + 3 < NSTEP) af[s_ % 3] = read_a(s_ + 3);
      if (kw < 2) {
        if constexpr (MF == 8) {
          if (mf >= 2) wf[(kw + 1) & 1][mf - 2] = read_w(kw + 1, mf - 2);
        } else {  // four steps per kw: two W fragments per step during the first three
          if (mf < 3) {
            wf[(kw + 1) & 1][2 * mf] = read_w(kw + 1, 2 * mf);
            wf[(kw + 1) & 1][2 * mf + 1] = read_w(kw + 1, 2 * mf + 1);
          }
        }
      }
    }
    slot = slot == 2 ? 0 : slot + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 7" ::: "memory");  // (the repeated pieces of the last sub-stage; the last products -> v_accvgpr_read)
#pragma unroll
  for (int mf = 0; mf < MF; ++mf)
#pragma unroll
    for (int n = 0; n < 6; ++n) asm volatile("" : "+a"(acc[mf][n]));

  // epilogue: lane (fr, fg) owns tile position 16 MF wave + 16 mf + fr, channels 16 n + 4 fg .. + 3
  f32x4 bv[6];
#pragma unroll
  for (int n = 0; n < 6; ++n) bv[n] = bias ? *reinterpret_cast<const f32x4*>(bias + 16 * n + 4 * fg) : f32x4{0.f, 0.f, 0.f, 0.f};
  const size_t shift = (size_t)Wp + 1;
  // gamma != nullptr: the NEXT layer's RMS_norm (+ SiLU) applied to the bf16-rounded result (ce_conv3d_gemm_rms_silu_bf16).  The four lanes fg
  // of a position hold its 96 channels - sum of squares in registers, two lane exchanges; the formula of rms_silu_kernel on the values it
  // would have read.  out2 == nullptr: only the normalised activation is stored (the first conv of a ResidualBlock feeds nothing but the
  // block's second norm, wan2pt1.py:195-200); out2 != nullptr: the result itself to `out` (the next block's shortcut) AND its normalised
  // form to `out2` (the next block's first norm) - the pass that would have re-read it is gone.
  f32x4 gv[6];
  if (gamma != nullptr) {
#pragma unroll
    for (int n = 0; n < 6; ++n) gv[n] = *reinterpret_cast<const f32x4*>(gamma + 16 * n + 4 * fg);
  }
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) {
    const int pt = 16 * MF * wave + 16 * mf + fr;
    const long long r = r0 + pt;
    const bool valid = pt < CR_TM && r < r_end;
    const size_t off = ((size_t)(valid ? r : 0) + shift) * ocs + 4 * fg;  // (invalid lanes: position 0 - in range for the residual loads; never stored)
    u32x2 rr[6];
    if (res) {  // (the six loads of a position issued together)
#pragma unroll
      for (int n = 0; n < 6; ++n) rr[n] = *reinterpret_cast<const u32x2*>(res + off + 16 * n);
    }
    u32x2 pk[6];
    float ss = 0.f;
#pragma unroll
    for (int n = 0; n < 6; ++n) {
      const f32x4 v4 = acc[mf][n] + bv[n];
      pk[n] = u32x2{pack_bf16(v4[0], v4[1]), pack_bf16(v4[2], v4[3])};
      if (res)  // bf16(res + bf16(acc + bias)): the rounding of the GEMM route's residual epilogue
        pk[n] = u32x2{pack_bf16(bf16lo(pk[n][0]) + bf16lo(rr[n][0]), bf16hi(pk[n][0]) + bf16hi(rr[n][0])),
                      pack_bf16(bf16lo(pk[n][1]) + bf16lo(rr[n][1]), bf16hi(pk[n][1]) + bf16hi(rr[n][1]))};
      ss += bf16lo(pk[n][0]) * bf16lo(pk[n][0]) + bf16hi(pk[n][0]) * bf16hi(pk[n][0]) + bf16lo(pk[n][1]) * bf16lo(pk[n][1]) +
            bf16hi(pk[n][1]) * bf16hi(pk[n][1]);
    }
    if (gamma == nullptr || out2 != nullptr) {
      if (valid) {
#pragma unroll
        for (int n = 0; n < 6; ++n) *reinterpret_cast<u32x2*>(out + off + 16 * n) = pk[n];
      }
    }
    if (gamma != nullptr) {  // (wave-uniform)
      ss += __shfl_xor(ss, 16, 64);
      ss += __shfl_xor(ss, 32, 64);
      const float scale = sqrtf(96.0f) / fmaxf(sqrtf(ss), 1e-12f);
      bf16* const nrow = (out2 != nullptr ? out2 : out) + off;
      if (valid) {
#pragma unroll
        for (int n = 0; n < 6; ++n) {
          float y0 = bf16lo(pk[n][0]) * scale * gv[n][0], y1 = bf16hi(pk[n][0]) * scale * gv[n][1];
          float y2 = bf16lo(pk[n][1]) * scale * gv[n][2], y3 = bf16hi(pk[n][1]) * scale * gv[n][3];
          if (apply_silu) y0 = silu_fast(y0), y1 = silu_fast(y1), y2 = silu_fast(y2), y3 = silu_fast(y3);
          *reinterpret_cast<u32x2*>(nrow + 16 * n) = u32x2{pack_bf16(y0, y1), pack_bf16(y2, y3)};
        }
      }
    }
  }
}

template <int NC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv3x3_c96_kernel(
    const bf16* __restrict__ in, long long in_bytes, const bf16* __restrict__ wgt, int ldw, const float* __restrict__ bias, bf16* __restrict__ out,
    const bf16* __restrict__ res, int rows, int KT, int Wp, int FS, int S, int ocs, int T_out, int tiles_per_frame, const float* __restrict__ gamma,
    int apply_silu, bf16* __restrict__ out2) {
  conv3x3_c96_body<NC, 4>(in, in_bytes, wgt, ldw, bias, out, res, rows, KT, Wp, FS, S, ocs, T_out, tiles_per_frame, gamma, apply_silu, out2);
}

template <int NC>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_c96_w8_kernel(
    const bf16* __restrict__ in, long long in_bytes, const bf16* __restrict__ wgt, int ldw, const float* __restrict__ bias, bf16* __restrict__ out,
    const bf16* __restrict__ res, int rows, int KT, int Wp, int FS, int S, int ocs, int T_out, int tiles_per_frame, const float* __restrict__ gamma,
    int apply_silu, bf16* __restrict__ out2) {
  conv3x3_c96_body<NC, 8>(in, in_bytes, wgt, ldw, bias, out, res, rows, KT, Wp, FS, S, ocs, T_out, tiles_per_frame, gamma, apply_silu, out2);
}

}  // namespace

extern "C" int ce_gemm256w4_seg2_launch(const void* A, const void* W, void* C, const float* bias, int epilogue, const void* res, int M, int N,
                                        int K, int lda, int ldw, int ldc, int ldres, int a_seg_k, long long a_seg_stride, int a_seg2_k,
                                        long long a_seg2_stride, int n_tile, hipStream_t stream);

static int conv3d_gemm_launch(const void* in_stack, const void* weight, int ldw, const float* bias, void* out_stack, const void* res_stack,
                              int T_out, int H, int W, int Cin, int Cout, int KT, int out_cstride, int n_tile, const float* gamma, int apply_silu,
                              void* normed_stack, hipStream_t stream) {
  if (!in_stack || !weight || !out_stack) return CE_ERR_ARG;
  if (T_out <= 0 || H <= 0 || W <= 0 || (KT != 1 && KT != 3) || (n_tile != 0 && n_tile != 1 && n_tile != 2 && n_tile != 96 && n_tile != 128 && n_tile != 256)) return CE_ERR_SHAPE;
  if ((Cin % 32) || (Cout & 7) || (out_cstride & 7) || out_cstride < Cout) return CE_ERR_SHAPE;
  const int Hp = H + 2, Wp = W + 2;
  // one (kt, kh) run of the K axis = the 3 Cin channels of three neighbouring pixels, rounded up to whole 64-wide K-tiles (Cin = 96: 288
  // -> 320; the 32 surplus columns read the next pixel's first channels against zero weights)
  const int seg = (3 * Cin + 63) / 64 * 64;
  const int ktiles = KT * 3 * (seg / 64), kpad = (ktiles + 1) / 2 * 2 * 64;
  if (ldw < kpad || (ldw & 7)) return CE_ERR_SHAPE;
  const long long rows = (long long)T_out * Hp * Wp - 2ll * (Wp + 1);
  if (rows <= 0 || rows * Cin * 2 >= (1ll << 32) || rows * out_cstride * 2 >= (1ll << 32)) return CE_ERR_SHAPE;
  if ((n_tile == 1 || n_tile == 2) && !(Cout == 96 && (Cin == 32 || Cin == 96 || Cin == 192))) return CE_ERR_SHAPE;
  const bool slab_shape = Cout == 96 && (Cin == 32 || Cin == 96 || Cin == 192) && 2ll * Hp * Wp * Cin * 2 + 3ll * Wp * Cin * 2 < (1ll << 31);
  if (gamma != nullptr && (!slab_shape || (n_tile != 0 && n_tile != 1 && n_tile != 2))) return CE_ERR_SHAPE;  // (the fused norm exists on the slab kernel only)
  if ((n_tile == 0 || n_tile == 1 || n_tile == 2) && Cout == 96 && (Cin == 32 || Cin == 96 || Cin == 192) && 2ll * Hp * Wp * Cin * 2 + 3ll * Wp * Cin * 2 < (1ll << 31)) {
    // the 96-channel full-resolution layers: the input slab itself in the LDS, 512 positions x 96 channels per workgroup (conv3x3_c96_kernel)
    static bool done_[CE_MAX_DEVICES] = {};
    bool& done = done_[ce_device_slot()];
    if (!done) {
      (void)hipFuncSetAttribute((const void*)conv3x3_c96_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, CR_SMEM);
      (void)hipFuncSetAttribute((const void*)conv3x3_c96_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, CR_SMEM);
      (void)hipFuncSetAttribute((const void*)conv3x3_c96_w8_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, CR_SMEM);
      (void)hipFuncSetAttribute((const void*)conv3x3_c96_w8_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, CR_SMEM);
      (void)hipFuncSetAttribute((const void*)conv3x3_c96_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, CR_SMEM);
      (void)hipFuncSetAttribute((const void*)conv3x3_c96_w8_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, CR_SMEM);
      done = true;
    }
    const long long in_bytes = (long long)(T_out + KT) * Hp * Wp * Cin * 2;  // T_out + KT - 1 frames and the slack frame
    const int tiles_per_frame = (Hp * Wp + CR_TM - 1) / CR_TM;
    const dim3 grid((unsigned)(((long long)tiles_per_frame * T_out + 7) / 8 * 8));
#define CE_C96(KERNEL, THREADS)                                                                                                               \
  hipLaunchKernelGGL(KERNEL, grid, dim3(THREADS), CR_SMEM, stream, (const bf16*)in_stack, in_bytes, (const bf16*)weight, ldw, bias, (bf16*)out_stack, \
                     (const bf16*)res_stack, (int)rows, KT, Wp, Hp * Wp, seg, out_cstride, T_out, tiles_per_frame, gamma, apply_silu, (bf16*)normed_stack)
    if (n_tile == 2) {  // (A/B partner: one wave per SIMD)
      if (Cin == 96) CE_C96(conv3x3_c96_kernel<3>, 256);
      else if (Cin == 192) CE_C96(conv3x3_c96_kernel<6>, 256);
      else CE_C96(conv3x3_c96_kernel<1>, 256);
    } else {
      if (Cin == 96) CE_C96(conv3x3_c96_w8_kernel<3>, 512);
      else if (Cin == 192) CE_C96(conv3x3_c96_w8_kernel<6>, 512);
      else CE_C96(conv3x3_c96_w8_kernel<1>, 512);  // (the encoder's 3 -> 96 stem on its 32-channel padded frames: 27 k-steps where 3 would do, still 2 x the implicit GEMM)
    }
#undef CE_C96
    const long long nb = (long long)T_out * (2 * Wp + 2 * (Hp - 2)) * (Cout / 8);
    hipLaunchKernelGGL(zero_border_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, stream, (bf16*)out_stack, T_out, Hp, Wp, Cout / 8,
                       out_cstride / 8);
    if (normed_stack != nullptr)
      hipLaunchKernelGGL(zero_border_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, stream, (bf16*)normed_stack, T_out, Hp, Wp, Cout / 8,
                         out_cstride / 8);
    return (int)hipGetLastError();
  }
  if (n_tile == 0) {
    // Priced per macro tile from the A/B of the 720p decode shapes (profiles/r05_conv_gemm_tile96_ab.txt): a 256 x 96 / 256 x 128 / 256 x 256 tile
    // costs 0.82 : 1 : 1.6 (the narrower ones read more LDS bytes per MFMA); a grid of a few rounds of 256 CUs pays whole rounds.  96 output
    // channels -> the 96-wide tile (a 128-wide one idles a quarter of every MFMA); 384 channels at 90 x 160 -> 96 too (four tile columns
    // fill the chip where three leave a third of it idle); the big 192 / 384-channel layers stay on 256 / 128.
    const long long tiles_m = (rows + 255) / 256;
    const int widths[3] = {96, 128, 256};
    const double cost[3] = {0.82, 1.0, 1.6};
    double best = 0.0;
    for (int i = 0; i < 3; ++i) {
      const long long tiles = tiles_m * ((Cout + widths[i] - 1) / widths[i]);
      const double rounds = tiles >= 1024 ? (double)tiles / 256.0 : (double)((tiles + 255) / 256);
      const double t = rounds * cost[i];
      if (n_tile == 0 || t < best) {
        best = t;
        n_tile = widths[i];
      }
    }
  }
  const size_t shift = (size_t)(Wp + 1);
  bf16* c0 = (bf16*)out_stack + shift * out_cstride;
  const bf16* r0 = res_stack ? (const bf16*)res_stack + shift * out_cstride : nullptr;
  const int rc = ce_gemm256w4_seg2_launch(in_stack, weight, c0, bias, res_stack ? 2 /* EPI_GATE_RES, no gate: bf16(res + bf16(acc + bias)) */ : 0,
                                          r0, (int)rows, Cout, kpad, Cin, ldw, out_cstride, out_cstride, seg, (long long)Wp * Cin, 3 * seg,
                                          (long long)Hp * Wp * Cin, n_tile, stream);
  if (rc != CE_OK) return rc;
  const long long n = (long long)T_out * (2 * Wp + 2 * (Hp - 2)) * (Cout / 8);
  hipLaunchKernelGGL(zero_border_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (bf16*)out_stack, T_out, Hp, Wp, Cout / 8,
                     out_cstride / 8);
  return (int)hipGetLastError();
}

CE_API int ce_conv3d_gemm_bf16(const void* in_stack, const void* weight, int ldw, const float* bias, void* out_stack, const void* res_stack,
                                   int T_out, int H, int W, int Cin, int Cout, int KT, int out_cstride, int n_tile, hipStream_t stream) {
  return conv3d_gemm_launch(in_stack, weight, ldw, bias, out_stack, res_stack, T_out, H, W, Cin, Cout, KT, out_cstride, n_tile, nullptr, 0, nullptr, stream);
}

// The convolution of ce_conv3d_gemm_bf16 with the NEXT layer's RMS_norm (+ SiLU) in its epilogue (ce_rms_silu_bf16's formula on the bf16-rounded
// result y = bf16(conv(in) + bias) [then bf16(res + y)]) - Cout == 96 and Cin 32 / 96 / 192 only (the slab kernel, whose lanes hold all 96
// channels of a position):
//   out_stack == NULL: normed_stack = [silu](RMS_norm(y) gamma) and y itself is never written (no residual) - CausalConv3d -> RMS_norm -> SiLU
//                      INSIDE a ResidualBlock (wan2pt1.py:195-200);
//   out_stack != NULL: y to out_stack (the next block's shortcut operand) and its normalised form to normed_stack (the next block's first norm,
//                      or the head's): the pass that would have re-read y is gone.
CE_API int ce_conv3d_gemm_rms_silu_bf16(const void* in_stack, const void* weight, int ldw, const float* bias, void* out_stack, const void* res_stack,
                                            void* normed_stack, int T_out, int H, int W, int Cin, int Cout, int KT, int out_cstride, const float* gamma,
                                            int apply_silu, hipStream_t stream) {
  if (!gamma || !normed_stack || (!out_stack && res_stack)) return CE_ERR_ARG;
  if (out_stack == nullptr)  // (the kernel's single-output form writes the normalised activation through its `out`)
    return conv3d_gemm_launch(in_stack, weight, ldw, bias, normed_stack, nullptr, T_out, H, W, Cin, Cout, KT, out_cstride, 0, gamma, apply_silu, nullptr, stream);
  return conv3d_gemm_launch(in_stack, weight, ldw, bias, out_stack, res_stack, T_out, H, W, Cin, Cout, KT, out_cstride, 0, gamma, apply_silu, normed_stack,
                            stream);
}

CE_API int ce_rms_silu_bf16(const void* x, void* y, const float* gamma, long long npix, int C, int H, int W, int in_border,
                                int out_border, int apply_silu, hipStream_t stream) {
  if (!x || !y || !gamma || npix <= 0) return CE_ERR_ARG;
  if ((C & 7) || C > 512 || H <= 0 || W <= 0 || npix % ((long long)H * W)) return CE_ERR_SHAPE;
  const int chunks = C / 8;
  int lpa = 16;
  while (chunks % lpa) --lpa;  // active lanes per pixel: the largest divisor of the chunk count that fits a 16-lane group
  const int per = chunks / lpa;
  const long long rows = npix / W;  // frames x image rows
  if (rows > 0x7fffffffll) return CE_ERR_SHAPE;
#define CE_RMS(P, G)                                                                                                                \
  hipLaunchKernelGGL((rms_silu_kernel<P, G>), dim3((unsigned)rows, (unsigned)((W + 16 * G - 1) / (16 * G))), dim3(256), 0, stream,   \
                     (const bf16*)x, (bf16*)y, gamma, C, lpa, H, W, W + 2 * in_border, in_border, W + 2 * out_border, out_border,   \
                     apply_silu)
  switch (per) {
    case 1: CE_RMS(1, 4); break;
    case 2: CE_RMS(2, 4); break;
    case 3: CE_RMS(3, 2); break;
    case 4: CE_RMS(4, 2); break;
    default: return CE_ERR_SHAPE;  // (C = 8 * a prime > 16 ...: no VAE width)
  }
#undef CE_RMS
  return (int)hipGetLastError();
}

/* Zero the one-pixel border of T frames [H+2][W+2][ld] (channels [0, C)): what a producer that writes interiors only leaves to do on a
 * buffer that was not zero-filled. */
CE_API int ce_zero_border_bf16(void* frames, int T, int H, int W, int C, int ld, hipStream_t stream) {
  if (!frames || T <= 0 || H <= 0 || W <= 0 || (C & 7) || (ld & 7) || ld < C) return CE_ERR_ARG;
  const int Hp = H + 2, Wp = W + 2;
  const long long n = (long long)T * (2 * Wp + 2 * (Hp - 2)) * (C / 8);
  hipLaunchKernelGGL(zero_border_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (bf16*)frames, T, Hp, Wp, C / 8, ld / 8);
  return (int)hipGetLastError();
}

CE_API int ce_upsample2x_bf16(const void* x, void* y, int T, int C, int H, int W, hipStream_t stream) {
  if (!x || !y || (C & 7) || T <= 0) return CE_ERR_ARG;
  const long long n = (long long)T * (2 * H) * (2 * W) * (C / 8);
  hipLaunchKernelGGL(upsample2x_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const bf16*)x, (bf16*)y, n, C / 8,
                     H, W, W + 2, 2 * W + 2);
  return (int)hipGetLastError();
}

CE_API int ce_softmax_rows_f32_bf16(const float* scores, void* probs, int M, int n, int npad, int ld, int ldp, float scale,
                                        hipStream_t stream) {
  if (!scores || !probs || M <= 0 || n <= 0 || npad < n || npad > ldp) return CE_ERR_ARG;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(M), dim3(256), 0, stream, scores, (bf16*)probs, n, npad, ld, ldp, scale);
  return (int)hipGetLastError();
}

// ---- mid-block attention of the Wan VAE as ONE flash-style kernel (wan2pt1.py:223-259: a single head over the h*w positions of a
// frame, head dim = C = 384 at the shipped width) -----------------------------------------------------------------------------------
// No [HW, HW] score matrix in memory (0.83 GB fp32 per frame at 720p, 2.7 GB at 1584x1056): scores, online softmax and P.V stay in
// registers, v_mfma_f32_16x16x32_bf16:
//   S^T block = K.Q^T   (first operand K: lane (fr, fg) then owns query fr and the four K-image rows 16 kb + 4 fg + r of row block kb)
//   O^T block = V^T.P^T (first operand V^T rows = output channels; the P operand is taken straight from the S registers)
// so a lane keeps one query's statistics (running maximum, partial row sum) and rescales its own accumulator values.
// Round 6 (rounds 3-5: 64-row workgroups, register-staged 64-key tiles, 0.13 of the matrix peak - every workgroup pulled the whole K and
// V^T through its CU for 64 query rows, ~10 B / cycle / CU of L2 -> LDS traffic was the bound, not the LDS reads the round-5 notes blamed):
//   * a wave owns RB = 2 blocks of 16 query rows (128 per workgroup: half the bytes per flop); their Q fragments live in registers
//     (2 x C / 32 x 4 = 96 at C = 384; one wave per SIMD owns 512) and every K / V^T fragment read feeds two MFMAs;
//   * K and V^T tiles of 32 keys arrive by LDS-DMA (buffer_load ... lds, 1 KiB per wave instruction) into a ring of three (K, V^T) slots,
//     TWO tiles ahead of the products, one barrier per tile; no staging registers, no ds_write.  Bank conflicts are taken out on the DMA's
//     SOURCE side:
//       K image  [C / 128 column groups][32 rows][16 chunks of 16 B]: position p of image row rho holds chunk p ^ (rho & 15) - a
//                ds_read_b128 of 16 rows at one chunk index hits 16 distinct bank quads in each of the instruction's four lane groups;
//                image row rho holds key 8 (rho >> 2 & 3) + 4 (rho >> 4) + (rho & 3) of the tile, so that the eight P values a lane
//                owns after S^T (rows 4 fg + r and 16 + 4 fg + r) are the keys 8 fg .. 8 fg + 7 = one MFMA k-slot group in natural order;
//       V^T image [C rows][4 chunks]: keys in natural order, position p of row d holds chunk p ^ (3 (d >> 3 & 1)): ONE ds_read_b128 per
//                fragment, conflict-free in the same four lane groups;
//   * the running maximum moves only when a tile's maximum exceeds it by more than 8 (in the exp2 domain: P <= 256, exact after the
//     normalisation): the 192 accumulators are rescaled in the first tile and practically never again;
//   * 128-row workgroups are only 113 at 14 400 positions, so the KEY axis is split over gridDim.y workgroups (2 at 720p: 226 workgroups
//     on 256 CUs): each writes its un-normalised fp32 O and its (maximum, row sum) to the caller's workspace and
//     attn_1head_merge_kernel combines them - O = sum_s O_s 2^(m_s - m) / sum_s l_s 2^(m_s - m); one split writes bf16 O directly.
namespace {

template <int C>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void attn_1head_kernel(
    const bf16* __restrict__ Q, const bf16* __restrict__ K, const bf16* __restrict__ Vt, bf16* __restrict__ O, int Nq, int Nk, int ldq, int ldk,
    int ldvt, int ldo, float sl2, float* __restrict__ wsO, float* __restrict__ wsML, int tiles_per_split) {
  constexpr int KS = C / 32;            // k-steps of the score product
  constexpr int DB = C / 16;            // 16-channel blocks of the output
  constexpr int CG = C / 128;           // 128-channel column groups of the K image
  constexpr int RB = 2;                 // 16-row query blocks per wave
  constexpr int KT = 32;                // keys per tile
  constexpr int K_TILE = KT * C * 2;    // 24 KiB at C = 384
  constexpr int V_TILE = C * KT * 2;
  constexpr int SLOT = K_TILE + V_TILE;
  constexpr int NVD = C / 64;           // V^T DMA pieces per wave and tile (the K image: 2 CG - the same number)
  constexpr int NDMA = 2 * CG + NVD;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int q0 = blockIdx.x * (64 * RB) + wave * (16 * RB);
  const int ntiles = (Nk + KT - 1) / KT;
  const int t_begin = blockIdx.y * tiles_per_split, t_end = min(ntiles, t_begin + tiles_per_split);
  const int nt = max(t_end - t_begin, 0);

  // LDS-DMA: rows past Nk read as zero (their scores are masked below; V^T columns past Nk are zero by contract)
  const auto k_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)K, 0, (uint32_t)(Nk - 1) * (uint32_t)(ldk * 2) + C * 2u, 0x00020000);
  const auto v_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Vt, 0, (uint32_t)(C - 1) * (uint32_t)(ldvt * 2) + (uint32_t)ntiles * (KT * 2u), 0x00020000);
  // K piece (cg, j) of wave w = image rows 4 (w + 4 j) + fg of column group cg, position fr <- chunk fr ^ (4 w + fg) of key 8 w + 4 j + fg
  const int k_voff = (8 * wave + fg) * ldk * 2 + ((fr ^ (4 * wave + fg)) << 4), k_step = 4 * ldk * 2, k_tile_step = KT * ldk * 2;
  // V^T piece j of wave w = rows 16 (w + 4 j) + (lane >> 2), position lane & 3 <- chunk (lane & 3) ^ (3 (row >> 3 & 1))
  const int v_voff = (16 * wave + (lane >> 2)) * ldvt * 2 + (((lane & 3) ^ (3 * ((lane >> 5) & 1))) << 4), v_step = 64 * ldvt * 2;
  auto dma = [&](int t, int slot) __attribute__((always_inline)) {
    unsigned char* const sb = smem + slot * SLOT;
#pragma unroll
    for (int cg = 0; cg < CG; ++cg)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        // (the whole key offset in the VGPR operand: the range check that zeroes the rows past Nk covers the VGPR offset)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(k_rsrc, (lds_void_c*)(sb + cg * 8192 + (wave + 4 * j) * 1024), 16,
                                                 k_voff + t * k_tile_step + j * k_step + cg * 256, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NVD; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(v_rsrc, (lds_void_c*)(sb + K_TILE + (wave + 4 * j) * 1024), 16, v_voff, t * (KT * 2) + j * v_step, 0, 0);
  };
  // Q fragments of this wave's 2 x 16 rows, read once (rows past Nq: clamped copies, never stored); then tiles 0 and 1 of this split
  bf16x8 qf[RB][KS];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      qf[rb][ks] = *reinterpret_cast<const bf16x8*>(Q + (size_t)min(q0 + 16 * rb + fr, Nq - 1) * ldq + 32 * ks + 8 * fg);
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {  // (retires the Q loads: the DMA queue below is counted)
      if (rb == 0) asm volatile("" : "+a"(qf[rb][ks]));  // the first block's fragments beside the 192 accumulators (240 of 256), ...
      else asm volatile("" : "+v"(qf[rb][ks]));          // ... the second block's in the vector file
    }
  __builtin_amdgcn_sched_barrier(0);
  dma(min(t_begin, ntiles - 1), 0);
  __builtin_amdgcn_sched_barrier(0);
  dma(min(t_begin + 1, ntiles - 1), 1);
  __builtin_amdgcn_sched_barrier(0);

  // fragment read offsets inside a slot: K (kb, ks) at koff[ks & 3] + (ks >> 2) 8192 + kb 4096; V^T (db) at voff + db 1024
  int koff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) koff[i] = fr * 256 + (((4 * i + fg) ^ fr) << 4);
  const int voff = K_TILE + fr * 64 + ((fg ^ (3 * ((fr >> 3) & 1))) << 4);

  f32x4 o[RB][DB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int d = 0; d < DB; ++d) {
      o[rb][d] = f32x4{0.f, 0.f, 0.f, 0.f};
      asm volatile("" : "+a"(o[rb][d]));
    }
  float m_run[RB], l_run[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) m_run[rb] = -3.0e38f, l_run[rb] = 0.f;

  int slot = 0;
  for (int i = 0; i < nt; ++i) {
    const int t = t_begin + i;
    // this wave's pieces of tile i have landed (the NDMA of tile i + 1 may fly); behind the barrier everybody's have, and everybody is
    // done with tile i - 1, whose slot takes tile i + 2
    if constexpr (NDMA == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    static_assert(NDMA == 12 || NDMA == 4, "vmcnt immediates above");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    dma(min(t + 2, ntiles - 1), slot == 0 ? 2 : slot - 1);  // (past the end: the last tile again - harmless, and the loop stays branch-free)
    __builtin_amdgcn_sched_barrier(0);
    const unsigned char* const sb = smem + slot * SLOT;
    // The matrix instructions are inline asm with their register files spelled out - accumulators of O ("+a") and the first query block's
    // fragments ("a") in the accumulator file, the second block's, the scores and the streamed fragments in the vector file: left to hipcc
    // (builtins) 192 + 96 long-lived registers ended with seven Q fragments in scratch, re-read every tile behind a vmcnt(0) each (which
    // also drains the DMA queue).  hipcc moves no load across a volatile asm, so the fragment reads are pipelined by hand: a ring of PRE
    // fragments runs ahead of the products (an LDS read returns in ~100-130 cycles = 3-4 steps of two 16-cycle MFMAs).
    constexpr int PRE = 4;
    auto read_k = [&](int n) __attribute__((always_inline)) {  // step n = (ks, kb) = (n >> 1, n & 1)
      return *reinterpret_cast<const bf16x8*>(sb + koff[(n >> 1) & 3] + (n >> 3) * 8192 + (n & 1) * 4096);
    };
    auto read_v = [&](int d) __attribute__((always_inline)) { return *reinterpret_cast<const bf16x8*>(sb + voff + d * 1024); };
    bf16x8 fr_[PRE];
#pragma unroll
    for (int n = 0; n < PRE; ++n) fr_[n] = read_k(n);
    f32x4 s[RB][2];
#pragma unroll
    for (int n = 0; n < 2 * KS; ++n) {
      const int ks = n >> 1, kb = n & 1;
      const bf16x8 kf = fr_[n % PRE];
      if (ks == 0) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(s[0][kb]) : "v"(kf), "a"(qf[0][ks]));
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(s[1][kb]) : "v"(kf), "v"(qf[1][ks]));
      } else {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(s[0][kb]) : "v"(kf), "a"(qf[0][ks]));
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(s[1][kb]) : "v"(kf), "v"(qf[1][ks]));
      }
      if (n + PRE < 2 * KS) fr_[n % PRE] = read_k(n + PRE);
      else fr_[n % PRE] = read_v(n + PRE - 2 * KS);  // the first V^T fragments fly under the softmax
    }
    asm volatile("s_nop 7" : "+v"(s[0][0]), "+v"(s[0][1]), "+v"(s[1][0]), "+v"(s[1][1]));  // 4-pass results -> the vector ALU: 7 wait states
    // online softmax in the exp2 domain, per query block; s[rb][kb][r] belongs to key 32 t + 8 fg + 4 kb + r
    if (t == ntiles - 1 && (Nk & (KT - 1))) {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (t * KT + 8 * fg + 4 * kb + r >= Nk) s[rb][kb][r] = -3.0e38f;
    }
    bf16x8 pf[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      float mt = fmaxf(fmaxf(fmaxf(s[rb][0][0], s[rb][0][1]), fmaxf(s[rb][0][2], s[rb][0][3])),
                       fmaxf(fmaxf(s[rb][1][0], s[rb][1][1]), fmaxf(s[rb][1][2], s[rb][1][3])));
      mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
      mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
      mt *= sl2;  // (sl2 > 0)
      const bool move = mt > m_run[rb] + 8.0f;  // the same answer in the four lanes of a query
      if (__any(move)) {
        const float m_new = move ? mt : m_run[rb];
        const float alpha = __builtin_amdgcn_exp2f(m_run[rb] - m_new);
        m_run[rb] = m_new;
        l_run[rb] *= alpha;
#pragma unroll
        for (int d = 0; d < DB; ++d) {  // out of the accumulator file, scaled, put back - ONE accumulator at a time (the asm pins order
          asm volatile("" : "+a"(o[rb][d]));  // the reads: 96 values in flight at once cost the vector file the second block's Q)
          f32x4 x = o[rb][d];
          x *= alpha;
          o[rb][d] = x;
          asm volatile("" : "+a"(o[rb][d]));
        }
      }
      const float nm = -m_run[rb];
      float psum = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s[rb][kb][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[rb][kb][r], sl2, nm));
          psum += s[rb][kb][r];
        }
      l_run[rb] += psum;
      const u32x4 pw = {pack_bf16(s[rb][0][0], s[rb][0][1]), pack_bf16(s[rb][0][2], s[rb][0][3]), pack_bf16(s[rb][1][0], s[rb][1][1]),
                        pack_bf16(s[rb][1][2], s[rb][1][3])};
      pf[rb] = __builtin_bit_cast(bf16x8, pw);
    }
#pragma unroll
    for (int d = 0; d < DB; ++d) {
      const bf16x8 vf = fr_[(d + 2 * KS) % PRE];
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(o[rb][d]) : "v"(vf), "v"(pf[rb]));
      if (d + PRE < DB) fr_[(d + 2 * KS) % PRE] = read_v(d + PRE);
    }
    slot = slot == 2 ? 0 : slot + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 7" ::: "memory");  // (the run-ahead pieces past this split's end; the last products -> v_accvgpr_read)
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int d = 0; d < DB; ++d) asm volatile("" : "+a"(o[rb][d]));
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    float l = l_run[rb];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const int q = q0 + rb * 16 + fr;
    if (q >= Nq) continue;
    if (wsO == nullptr) {  // one split: normalised bf16 output
      const float inv = 1.0f / l;
      bf16* orow = O + (size_t)q * ldo + 4 * fg;
#pragma unroll
      for (int d = 0; d < DB; ++d) {
        const u32x2 pk = {pack_bf16(o[rb][d][0] * inv, o[rb][d][1] * inv), pack_bf16(o[rb][d][2] * inv, o[rb][d][3] * inv)};
        *reinterpret_cast<u32x2*>(orow + d * 16) = pk;
      }
    } else {  // a key split: un-normalised fp32 O [split][Nq][C] and (m, l) [split][Nq][2] for attn_1head_merge_kernel
      float* orow = wsO + ((size_t)blockIdx.y * Nq + q) * C + 4 * fg;
#pragma unroll
      for (int d = 0; d < DB; ++d) *reinterpret_cast<f32x4*>(orow + d * 16) = o[rb][d];
      if (fg == 0) {
        wsML[((size_t)blockIdx.y * Nq + q) * 2 + 0] = m_run[rb];
        wsML[((size_t)blockIdx.y * Nq + q) * 2 + 1] = l;
      }
    }
  }
}

// O[q][c] = sum_s O_s[q][c] 2^(m_s - m) / sum_s l_s 2^(m_s - m), m = max_s m_s: one thread per (query, 4 channels)
__global__ __launch_bounds__(256) void attn_1head_merge_kernel(const float* __restrict__ wsO, const float* __restrict__ wsML, bf16* __restrict__ O, int Nq,
                                                              int C, int ldo, int nsplit) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const int c4 = C / 4;
  if (idx >= (long long)Nq * c4) return;
  const int q = (int)(idx / c4), c = (int)(idx - (long long)q * c4) * 4;
  float m = -3.0e38f;
  for (int s_ = 0; s_ < nsplit; ++s_) m = fmaxf(m, wsML[((size_t)s_ * Nq + q) * 2]);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float l = 0.f;
  for (int s_ = 0; s_ < nsplit; ++s_) {
    const float w = __builtin_amdgcn_exp2f(wsML[((size_t)s_ * Nq + q) * 2] - m);
    l += wsML[((size_t)s_ * Nq + q) * 2 + 1] * w;
    const f32x4 v = *reinterpret_cast<const f32x4*>(wsO + ((size_t)s_ * Nq + q) * C + c);
    acc += v * w;
  }
  const float inv = 1.0f / l;
  const u32x2 pk = {pack_bf16(acc[0] * inv, acc[1] * inv), pack_bf16(acc[2] * inv, acc[3] * inv)};
  *reinterpret_cast<u32x2*>(O + (size_t)q * ldo + c) = pk;
}

}  // namespace

// O [Nq][ldo] = softmax(Q K^T * scale) V for ONE head of dimension C (128 or 384): Q [Nq][ldq], K [Nk][ldk] bf16 rows, Vt [C][ldvt] =
// V transposed (keys contiguous; columns [Nk, 64 ceil(Nk / 64)) must be finite - zero them).  Replaces the q.k^T / softmax / .v of the
// VAE's AttentionBlock (chronoedit/_src/tokenizers/wan2pt1.py:247-255, F.scaled_dot_product_attention on [b t, 1, h w, c]).
CE_API int ce_attention_1head_bf16(const void* Q, const void* K, const void* Vt, void* O, int Nq, int Nk, int C, int ldq, int ldk, int ldvt,
                                       int ldo, float softmax_scale, void* ws, long long ws_bytes, hipStream_t stream) {
  if (!Q || !K || !Vt || !O || Nq <= 0 || Nk <= 0) return CE_ERR_ARG;
  if ((C != 128 && C != 384) || (ldq & 7) || (ldk & 7) || (ldvt & 7) || (ldo & 3) || ldvt < (Nk + 63) / 64 * 64) return CE_ERR_SHAPE;
  const float sl2 = softmax_scale * 1.4426950408889634f;
  const int blocks = (Nq + 127) / 128, ntiles = (Nk + 31) / 32;  // 32-key tiles
  int cus = 256, dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  // key split: as many splits (<= 4, >= 16 key tiles each) as fill the chip, if the caller's workspace holds the fp32 partial results
  int nsplit = 1;
  if (ws != nullptr && blocks * 4 < cus * 3)
    for (int sp = 4; sp >= 2; --sp)
      if (blocks * sp <= cus + cus / 8 && ntiles >= 16 * sp && (long long)sp * Nq * (C + 2) * 4 <= ws_bytes) {
        nsplit = sp;
        break;
      }
  const int per = (ntiles + nsplit - 1) / nsplit;
  float* wsO = nsplit > 1 ? (float*)ws : nullptr;
  float* wsML = nsplit > 1 ? wsO + (size_t)nsplit * Nq * C : nullptr;
  const dim3 grid(blocks, nsplit), block(256);
  static bool done_[CE_MAX_DEVICES] = {};
  bool& done = done_[ce_device_slot()];
  const int smem384 = 3 * 2 * 32 * 384 * 2, smem128 = 3 * 2 * 32 * 128 * 2;  // ring of three (K, V^T) slots of 32 keys
  if (!done) {
    (void)hipFuncSetAttribute((const void*)attn_1head_kernel<384>, hipFuncAttributeMaxDynamicSharedMemorySize, smem384);
    (void)hipFuncSetAttribute((const void*)attn_1head_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, smem128);
    done = true;
  }
  if (C == 384)
    hipLaunchKernelGGL(attn_1head_kernel<384>, grid, block, smem384, stream, (const bf16*)Q, (const bf16*)K, (const bf16*)Vt, (bf16*)O, Nq, Nk, ldq,
                       ldk, ldvt, ldo, sl2, wsO, wsML, per);
  else
    hipLaunchKernelGGL(attn_1head_kernel<128>, grid, block, smem128, stream, (const bf16*)Q, (const bf16*)K, (const bf16*)Vt, (bf16*)O, Nq, Nk, ldq,
                       ldk, ldvt, ldo, sl2, wsO, wsML, per);
  if (nsplit > 1) {
    const long long n = (long long)Nq * (C / 4);
    hipLaunchKernelGGL(attn_1head_merge_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, wsO, wsML, (bf16*)O, Nq, C, ldo, nsplit);
  }
  return (int)hipGetLastError();
}
