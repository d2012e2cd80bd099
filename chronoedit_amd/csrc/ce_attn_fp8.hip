// MXFP8 self-attention for the fp8 mode of the DiT (BASELINE.json configs[4]: "fp8 weights+attn (CDNA4 fp8 MFMA)").
//
// The reference has no fp8 inference path, so the arithmetic CONTRACT is defined here and restated in
// oracle/dit_oracle.py::attention_mxfp8; what it replaces is F.scaled_dot_product_attention at transformer_chronoedit.py:91-104.
//   * Q, K (after RMSNorm + RoPE) and V are quantised to OCP MXFP8: e4m3 elements with one E8M0 (power-of-two) scale per block
//     of 32 consecutive elements ALONG THE CONTRACTION AXIS - the head channels d for Q and K, the keys for V.  Scale of a block =
//     2^(floor(log2 amax) - 8); elements = RNE(x / scale) clamped to +-448.
//   * S = Q.K^T and O = P.V run on v_mfma_scale_f32_32x32x64_f8f6f4 (block scales applied by the matrix pipe, fp32 accumulate).
//   * P = exp2((S - rowmax) * scale * log2e + 8) is rounded to e4m3 with unit scale (P <= 2^8, so nothing saturates and
//     probabilities down to 2^-17 of the row maximum survive); the row sum that normalises O is the fp32 sum of the un-rounded P.
//   * Online softmax over 64-key tiles with an exact running maximum; O / l rounded to bf16.
//
// Three kernels (all hand-written for gfx950, wave64):
//   rmsnorm_rope_mxfp8_kernel   K7+K8 of the bf16 path, writing MXFP8 q / k (+ scale bytes) instead of bf16: the quantisation is
//                               done ONCE per element by the producer, not once per consuming workgroup (29 of them at N = 7200).
//   v_mxfp8_transpose_kernel    V -> V^T tiles [head][d][key] in the key order the P operand has in the accumulator registers
//                               (so that NO cross-lane movement of P is needed), quantised per (d, 32-key block).
//   attn_fwd_mxfp8_kernel       8 waves x 32 query rows; K / V^T / scale tiles arrive by LDS-DMA (global_load_lds, no VGPR staging,
//                               no transposes in the loop), three stages, ONE barrier per 64-key tile; per tile and wave
//                               4 + 4 MFMAs of 32x32x64 (half the matrix-pipe time of the bf16 kernel) and 16 ds_read_b128.
// Bound: MX-fp8 MFMA (5 PFLOP/s dense); algorithmic flops = 4 * Nq * Nkv * 128 per head.
#include <type_traits>

#include "ce_common.h"

namespace {
#ifdef CE_DIAGNOSTICS
__device__ unsigned long long g_mx_exact_hits = 0ull;
#endif


typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(2))) short s16x2;
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

constexpr int HD = 128, QW = 32, KVB = 64;
constexpr int ROW_MAXC = 10;

// (mx_scale_byte / mx_inv_scale / clamp448: ce_common.h)

// ------------------------------------------------------------------------------------------------
// RMSNorm(across heads) [+ RoPE] -> MXFP8 (same rounding points as rmsnorm_rope_kernel up to the bf16 value, then quantised)
// ------------------------------------------------------------------------------------------------
template <bool FULL>
__global__ __launch_bounds__(256) void rmsnorm_rope_mxfp8_kernel(const bf16* __restrict__ x, const float* __restrict__ w,
                                                                 const float* __restrict__ cs, unsigned char* __restrict__ q8,
                                                                 unsigned char* __restrict__ sc, int M, int D, int ldx, int ldq,
                                                                 int head_dim, float eps, int rope_rows, float post_scale) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int cs_row = rope_rows > 0 ? row % rope_rows : row;
  const int nch = D >> 3;
  const bf16* xr = x + (size_t)row * ldx;
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
  unsigned char* qr = q8 + (size_t)row * ldq;
  unsigned char* sr = sc + (size_t)row * (D >> 5);
  // The weight and cos / sin vectors of a chunk are fetched ONE CHUNK AHEAD of its arithmetic (round 6: written as loads at their point of use,
  // hipcc waited for each chunk's four loads before anything else - ten serial L2 round trips per row: 3.3 TB/s where the bf16 kernel, whose
  // loop has no cross-lane step, runs 4.9), and the four scale bytes of 16 lanes leave as ONE dword store.
  // (a lane's chunks are 64 chunks = 512 channels apart: whenever 512 is a multiple of head_dim - 128 here - they all sit at the SAME place of
  //  their head and share one cos / sin entry: two loads per row instead of twenty - the row's 512-byte table was read forty times over)
  const bool cs_once = cs != nullptr && (512 % head_dim) == 0;
  f32x4 wv0[2], wv1[2], cv0[2], cv1[2];
  if (cs_once) {
    const float* p = cs + ((size_t)cs_row * half + (((lane * 8) % head_dim) >> 1)) * 2;
    cv0[0] = cv0[1] = *reinterpret_cast<const f32x4*>(p);
    cv1[0] = cv1[1] = *reinterpret_cast<const f32x4*>(p + 4);
  }
  auto fetch = [&](int i, int slot) __attribute__((always_inline)) {
    const int c = lane + 64 * i;
    if (FULL || c < nch) {
      wv0[slot] = *reinterpret_cast<const f32x4*>(w + c * 8);
      wv1[slot] = *reinterpret_cast<const f32x4*>(w + c * 8 + 4);
      if (cs != nullptr && !cs_once) {
        const int pair0 = ((c * 8) % head_dim) >> 1;
        const float* p = cs + ((size_t)cs_row * half + pair0) * 2;
        cv0[slot] = *reinterpret_cast<const f32x4*>(p);
        cv1[slot] = *reinterpret_cast<const f32x4*>(p + 4);
      }
    }
  };
  fetch(0, 0);
#pragma unroll
  for (int i = 0; i < ROW_MAXC; ++i) {
    const int c = lane + 64 * i;
    const bool on = FULL || c < nch;
    if (i + 1 < ROW_MAXC) fetch(i + 1, (i + 1) & 1);
    float v[8];
    float amax = 0.f;
    if (on) {
      const f32x4 w0 = wv0[i & 1], w1 = wv1[i & 1], cs0 = cv0[i & 1], cs1 = cv1[i & 1];
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
        // the value the bf16 path would have stored, times post_scale (q: softmax_scale * log2 e, so that Q.K^T comes out of the
        // matrix pipe in the exp2 domain and P is a bare v_exp_f32 per element; k: 1)
        v[2 * j] = round_bf16(v0) * post_scale;
        v[2 * j + 1] = round_bf16(v1) * post_scale;
        amax = fmaxf(amax, fmaxf(fabsf(v[2 * j]), fabsf(v[2 * j + 1])));
      }
    }
    // a 32-element block = 4 consecutive chunks = lanes 4a .. 4a+3
    amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
    amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
    const int byte = on ? mx_scale_byte(amax) : 0;
    // the scale bytes of four consecutive blocks (lanes 16 a, + 4, + 8, + 12) as one dword from lane 16 a
    int packed = byte | (__shfl_down(byte, 4, 64) << 8) | (__shfl_down(byte, 8, 64) << 16) | (__shfl_down(byte, 12, 64) << 24);
    if (on) {
      const float inv = mx_inv_scale(byte);
      int w0 = 0, w1 = 0;
      w0 = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(v[0] * inv), clamp448(v[1] * inv), w0, false);
      w0 = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(v[2] * inv), clamp448(v[3] * inv), w0, true);
      w1 = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(v[4] * inv), clamp448(v[5] * inv), w1, false);
      w1 = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(v[6] * inv), clamp448(v[7] * inv), w1, true);
      const u32x2 o = {(uint32_t)w0, (uint32_t)w1};
      *reinterpret_cast<u32x2*>(qr + c * 8) = o;
      if ((lane & 15) == 0) {
        if (FULL || c + 12 < nch) *reinterpret_cast<int*>(sr + (c >> 2)) = packed;  // (D % 128 == 0 whenever FULL; else byte by byte below)
        else
          for (int k = 0; k < 4 && c + 4 * k < nch; ++k) sr[(c >> 2) + k] = (unsigned char)(packed >> (8 * k));
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// V [tokens][ldv] (head h = columns h*128..) -> V^T tiles: v8t[(b*H + h)*128 + d][npad] bytes, sv[b][h][npad/64][d][2].
// Inside every 64-key tile the keys are stored in the order the P operand has in the S^T accumulator registers of the
// attention kernel: position p = 32 g + j  <->  key = 32 (j >> 4) + (j & 3) + 8 ((j & 15) >> 2) + 4 g   (g = lane half).
// One MX block = 32 CONSECUTIVE keys (sv[..][2 t + beta] = scale of keys 64 t + 32 beta ..).  Keys past the end are zeros.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int slot_key(int g, int j) { return 32 * (j >> 4) + (j & 3) + 8 * ((j & 15) >> 2) + 4 * g; }

__global__ __launch_bounds__(256) void v_mxfp8_transpose_kernel(const bf16* __restrict__ V, int ldv, unsigned char* __restrict__ v8t,
                                                                unsigned char* __restrict__ sv, int n_tokens, int H, int npad) {
  __shared__ __attribute__((aligned(16))) unsigned char tile[KVB * HD * 2];  // [64 keys][128 d] bf16
  const int t = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x;
  const bf16* vb = V + (size_t)b * n_tokens * ldv + h * HD;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (tid >> 4) + 16 * i, ck = tid & 15;
    const int kv = t * KVB + r;
    u32x4 val = {0u, 0u, 0u, 0u};
    if (kv < n_tokens) val = *reinterpret_cast<const u32x4*>(vb + (size_t)kv * ldv + ck * 8);
    *reinterpret_cast<u32x4*>(tile + r * (HD * 2) + ck * 16) = val;
  }
  __syncthreads();
  const int d = tid & 127, g = tid >> 7;
  // MX blocks of V are 32 CONSECUTIVE keys: in the P operand bytes 0-15 of both lane halves are keys 0-31 of the tile and bytes
  // 16-31 are keys 32-63 (slot_key), and the matrix unit takes the scale of byte half beta from lane (d, beta).  Thread (d, g)
  // holds the 32 keys of its lane half - 16 of each block - and meets its partner (d, 1 - g) through LDS for the block maxima.
  __shared__ float amx[2][2][HD];  // [lane half][key block][d]
  float v[32];
  float am0 = 0.f, am1 = 0.f;
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    v[j] = (float)*reinterpret_cast<const bf16*>(tile + slot_key(g, j) * (HD * 2) + d * 2);
    if (j < 16) am0 = fmaxf(am0, fabsf(v[j]));
    else am1 = fmaxf(am1, fabsf(v[j]));
  }
  amx[g][0][d] = am0;
  amx[g][1][d] = am1;
  __syncthreads();
  am0 = fmaxf(am0, amx[1 - g][0][d]);
  am1 = fmaxf(am1, amx[1 - g][1][d]);
  const int byte0 = mx_scale_byte(am0), byte1 = mx_scale_byte(am1);
  const float inv0 = mx_inv_scale(byte0), inv1 = mx_inv_scale(byte1);
  const int byte = g ? byte1 : byte0;  // this thread stores the scale of key block g
  uint32_t o[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float inv = i < 4 ? inv0 : inv1;  // positions j = 4 i .. 4 i + 3: j < 16 are keys of block 0, j >= 16 of block 1
    int wv = 0;
    wv = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(v[4 * i] * inv), clamp448(v[4 * i + 1] * inv), wv, false);
    wv = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(v[4 * i + 2] * inv), clamp448(v[4 * i + 3] * inv), wv, true);
    o[i] = (uint32_t)wv;
  }
  const size_t rowi = ((size_t)b * H + h) * HD + d;
  unsigned char* dst = v8t + rowi * npad + t * KVB + 32 * g;
  const u32x4 o0 = {o[0], o[1], o[2], o[3]}, o1 = {o[4], o[5], o[6], o[7]};
  *reinterpret_cast<u32x4*>(dst) = o0;
  *reinterpret_cast<u32x4*>(dst + 16) = o1;
  // scale bytes tile-major: sv[b][h][tile][d][beta] - one contiguous 256-B record per (head, tile), fetched by the attention kernel
  // with dword LDS-DMA (sub-dword LDS-DMA lanes land on a 4-byte pitch, so 2-byte pieces cannot be packed)
  sv[((((size_t)b * H + h) * (npad >> 6) + t) * HD + d) * 2 + g] = (unsigned char)byte;
}

// ------------------------------------------------------------------------------------------------
// attention
// ------------------------------------------------------------------------------------------------
constexpr int ST_K = 0, ST_V = 8192, ST_SK = 16384, ST_SV = 16640, STAGE = 17408;  // one stage: K8 8K | V8T 8K | sk 256 | sv 256 (+pad)
constexpr int NSTAGE = 3;
constexpr int OST_ROW = HD * 2 + 16;
constexpr int SMEM = NSTAGE * STAGE > 8 * QW * OST_ROW ? NSTAGE * STAGE : 8 * QW * OST_ROW;
constexpr float NEG_BIG = -1.0e30f;

__global__ __launch_bounds__(512, 2) void attn_fwd_mxfp8_kernel(const unsigned char* __restrict__ Q8, const unsigned char* __restrict__ SQ,
                                                               const unsigned char* __restrict__ K8, const unsigned char* __restrict__ SK,
                                                               const unsigned char* __restrict__ V8T, const unsigned char* __restrict__ SV,
                                                               bf16* __restrict__ O, int Nq, int Nkv, int npad, int H, int ldq8,
                                                               int ldk8, int ldo, int nqb) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr float scale_log2e = 1.0f;  // q arrives multiplied by softmax_scale * log2(e) (producer): S is in the exp2 domain
  const int D32 = (H * HD) >> 5;  // scale bytes per row of q / k
  {
    const size_t bz = blockIdx.y;
    Q8 += bz * Nq * ldq8;
    SQ += bz * Nq * D32;
    K8 += bz * Nkv * ldk8;
    SK += bz * Nkv * D32;
    V8T += bz * H * HD * npad;
    SV += bz * H * HD * (npad >> 5);  // [H][npad / 64][128][2]
    O += bz * Nq * ldo;
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  int head, qb;
  if ((H & 7) == 0) {
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    head = xcd + 8 * (local / nqb);
    qb = local % nqb;
  } else {
    head = blockIdx.x / nqb;
    qb = blockIdx.x % nqb;
  }
  const int q0 = qb * (QW * 8) + wave * QW;
  const int hoff = head * HD;
  const int ntiles = (Nkv + KVB - 1) / KVB;

  // Operand geometry of v_mfma_scale_f32_32x32x64_f8f6f4 (tools/probes/mx32_probe*.hip, profiles/r02_mx32_probe.txt): lane
  // (row r = l31, half g = hh) supplies 32 bytes; bytes 0-15 are k = 16 g .. 16 g + 15 and bytes 16-31 are k = 32 + 16 g .. of the
  // 64-deep step, i.e. MX block 0 (k < 32) is bytes 0-15 of BOTH halves and block 1 is bytes 16-31 of both; the scale byte
  // of block beta is taken from lane (r, beta).  So lane (r, g) fetches the 16-B chunks g and 2 + g of its row's 64 bytes and
  // passes the scale of block g.
  // Q^T B-operand
  i32x8 qf[2];
  int sqv[2];
  {
    const int qr = min(q0 + l31, Nq - 1);
    const unsigned char* qrow = Q8 + (size_t)qr * ldq8 + hoff + 16 * hh;
    const uint32_t sw = *reinterpret_cast<const uint32_t*>(SQ + (size_t)qr * D32 + head * 4);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const u32x4 a = *reinterpret_cast<const u32x4*>(qrow + 64 * ks), b = *reinterpret_cast<const u32x4*>(qrow + 64 * ks + 32);
      qf[ks] = i32x8{(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
      sqv[ks] = (int)((sw >> (16 * ks + 8 * hh)) & 0xffu);
    }
  }

  // per-lane DMA source offsets (the 16-B-chunk swizzles live on the SOURCE side: the LDS image of a DMA is lane-linear)
  const int k_row = 8 * wave + (lane >> 3);                       // K8 tile row this lane fetches (64 rows x 128 B)
  const int k_chunk = (lane & 7) ^ ((k_row >> 1) & 7);
  const int v_row = 16 * wave + (lane >> 2);                      // V8T tile row (128 d x 64 B)
  const int v_chunk = (lane & 3) ^ ((v_row >> 2) & 3);
  const unsigned char* vsrc = V8T + ((size_t)head * HD + v_row) * npad + v_chunk * 16;
  const unsigned char* svsrc = SV + (size_t)head * (npad >> 6) * 256 + 32 * wave + 4 * (lane & 7);  // + 256 per tile
  auto stage_tile = [&](int t, int slot) {  // tile t -> stage `slot`
    unsigned char* st = smem + slot * STAGE;
    const int kr = min(t * KVB + k_row, Nkv - 1);
    __builtin_amdgcn_global_load_lds((gbl_void*)(K8 + (size_t)kr * ldk8 + hoff + k_chunk * 16), (lds_void*)(st + ST_K + wave * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gbl_void*)(vsrc + (size_t)t * KVB), (lds_void*)(st + ST_V + wave * 1024), 16, 0, 0);
    if (lane < 8) {   // 8 key rows' scale words (4 B each) per wave
      const int sr = min(t * KVB + 8 * wave + lane, Nkv - 1);
      __builtin_amdgcn_global_load_lds((gbl_void*)(SK + (size_t)sr * D32 + head * 4), (lds_void*)(st + ST_SK + wave * 32), 4, 0, 0);
    }
    if (lane < 8)     // 16 d rows' scale pairs (8 dwords) per wave
      __builtin_amdgcn_global_load_lds((gbl_void*)(svsrc + (size_t)t * 256), (lds_void*)(st + ST_SV + wave * 32), 4, 0, 0);
  };

  f32x16 oacc[4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[m][r] = 0.f;
  float m_run = NEG_BIG, l_run = 0.f;

  stage_tile(0, 0);
  stage_tile(ntiles > 1 ? 1 : 0, 1);  // surplus prefetches re-fetch a valid tile into a free stage (their data is never read)

  for (int t = 0; t < ntiles; ++t) {
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // this wave's pieces of tile t have landed (tile t+1's four stay in flight)
    __builtin_amdgcn_s_barrier();                     // ... everybody's have, and everybody is done reading tile t-1's stage
    stage_tile(t + 2 < ntiles ? t + 2 : ntiles - 1, (t + 2) % NSTAGE);
    const unsigned char* st = smem + (t % NSTAGE) * STAGE;

    // ---- S^T = K . Q^T : two 32-key halves x two 64-deep k-steps
    f32x16 sacc[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[f][r] = 0.f;
      const int row = 32 * f + l31;
      const unsigned char* krow = st + ST_K + row * 128;
      const int sw = (row >> 1) & 7;
      const uint32_t skw = *reinterpret_cast<const uint32_t*>(st + ST_SK + row * 4);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int c0 = 4 * ks + hh;  // chunks g and 2 + g of this k-step
        const u32x4 a = *reinterpret_cast<const u32x4*>(krow + ((c0 ^ sw) << 4)), b = *reinterpret_cast<const u32x4*>(krow + (((c0 + 2) ^ sw) << 4));
        const i32x8 kf = {(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
        const int ska = (int)((skw >> (16 * ks + 8 * hh)) & 0xffu);
        sacc[f] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kf, qf[ks], sacc[f], 0, 0, 0, ska, 0, sqv[ks]);
      }
    }
    // ---- mask the key tail of the last tile: key = 64 t + 32 f + (r & 3) + 8 (r >> 2) + 4 hh
    if ((t + 1) * KVB > Nkv) {
      const int base = t * KVB + 4 * hh;
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (base + 32 * f + (r & 3) + 8 * (r >> 2) >= Nkv) sacc[f][r] = NEG_BIG;
    }
    // ---- online softmax: this lane's query row; the partner lane ^ 32 holds the other half of the keys
    float mx = sacc[0][0];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[f][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * scale_log2e);
    const float mc = m_new * scale_log2e - 8.0f;  // P = 2^8 exp2((S - m) c): uses the e4m3 range up to 256
    m_run = m_new;
    float psum = 0.f;
    i32x8 pf;
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float p[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          p[e] = __builtin_amdgcn_exp2f(fmaf(sacc[f][4 * i + e], scale_log2e, -mc));
          psum += p[e];
        }
        int wv = 0;
        wv = __builtin_amdgcn_cvt_pk_fp8_f32(p[0], p[1], wv, false);
        wv = __builtin_amdgcn_cvt_pk_fp8_f32(p[2], p[3], wv, true);
        pf[4 * f + i] = wv;  // k-slot j = 16 f + 4 i + e of lane half hh  <->  key 32 f + (r & 3) + 8 (r >> 2) + 4 hh, r = 4 i + e
      }
    l_run = l_run * alpha + psum;
    if (__any(alpha != 1.0f)) {
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[m][r] *= alpha;
    }
    // ---- O^T += V^T . P^T : four 32-d blocks, one 64-deep k-step each
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int dv = 32 * m + l31;
      const unsigned char* vrow = st + ST_V + dv * 64;
      const int sw = (dv >> 2) & 3;
      const u32x4 a = *reinterpret_cast<const u32x4*>(vrow + (((2 * hh) ^ sw) << 4)), b = *reinterpret_cast<const u32x4*>(vrow + (((2 * hh + 1) ^ sw) << 4));
      const i32x8 vf = {(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
      const int sva = (int)((*reinterpret_cast<const uint16_t*>(st + ST_SV + dv * 2) >> (8 * hh)) & 0xffu);
      oacc[m] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vf, pf, oacc[m], 0, 0, 0, sva, 0, 0x7f);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // surplus prefetches must land before the staging rows reuse the LDS
  __builtin_amdgcn_s_barrier();

  // ---- normalise, round to bf16, stage this wave's 32 x 128 tile and store whole 256-B rows
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  unsigned char* ost = smem + (size_t)(wave * QW + l31) * OST_ROW;
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const u32x2 val = {pack_bf16(oacc[m][4 * a + 0] * inv, oacc[m][4 * a + 1] * inv), pack_bf16(oacc[m][4 * a + 2] * inv, oacc[m][4 * a + 3] * inv)};
      *reinterpret_cast<u32x2*>(ost + (32 * m + 8 * a + 4 * hh) * 2) = val;
    }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = lane + 64 * i;
    const int rl = c >> 4, cc = c & 15;
    const int q = q0 + rl;
    if (q < Nq) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(smem + (size_t)(wave * QW + rl) * OST_ROW + cc * 16);
      *reinterpret_cast<u32x4*>(O + (size_t)q * ldo + hoff + cc * 8) = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Software-pipelined form (default).  The plain kernel above spends ~3000 cycles per 64-key tile and SIMD for 1024 cycles of
// matrix work: after halving the MFMA time the loop is VALU-bound (max, fma, exp2, sum, convert: ~180 VALU instructions per wave
// and tile) and nothing of it overlaps.  Here, per iteration and wave (the recipe of attn_fwd_sp_kernel in ce_attn.hip, adapted):
//     vmcnt / barrier / LDS-DMA of tile t+2
//   | S(t) = K(t).Q^T : 4 MFMAs whose accumulators START at -mc (the offset in use), so S arrives as "score - mc" in the exp2
//     domain (q is pre-multiplied by softmax_scale * log2 e by the producer) and P = exp2(S) is ONE v_exp_f32 per element;
//   | O += V^T(t-1).P^T(t-1) : 4 MFMAs, each followed by the v_exp and the e4m3 conversion of 8 elements of P(t) and the fetch of a
//     K(t+1) fragment;  then ONE more MFMA, ones.P(t), leaves the row sums of the rounded P(t) in every lane;
//   | speculative offset: no row maximum and no row-sum adds on the common path.  mc is set from the exact row maximum of tile 0 so
//     that P <= 2^3 there; later tiles are exponentiated against the same offset.  An element that outgrows e4m3 converts to NaN,
//     the NaN reaches the row sums, and one compare at the top of the next iteration sends that wave through the exact route:
//     S(t) again from the K tile still in LDS, offset moved to the true maximum, O and l rescaled, P(t) and its sums redone.
//     Nothing of a failed attempt survives (P(t) meets V(t) only in the next iteration, after the check).
// Five stages (LDS-DMA three tiles ahead; tile t-1 is still read while tile t+3 lands), one barrier per tile.
// ------------------------------------------------------------------------------------------------
constexpr int NSTAGE_SP = 5;

// acc += A.B as an asm statement: written with the builtin, hipcc SANK the four P.V MFMAs of a tile below the speculation check
// (their results are not needed until the next iteration), behind all of the softmax VALU work they are meant to run beside.
// A volatile asm stays where it is written.  "s_nop 1": a VALU-written scale register needs two wait states in front of the
// MFMA that reads it, and nothing inside an asm statement is padded by the compiler (cdna guide 5.7).
__device__ __forceinline__ void mfma_scale_acc_pinned(f32x16& acc, const i32x8& a, const i32x8& b, int sa, int sb) {
  asm volatile("s_nop 1\n\tv_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+v"(acc) : "v"(a), "v"(b), "v"(sa), "v"(sb));
}
constexpr int SMEM_SP = NSTAGE_SP * STAGE > 8 * QW * OST_ROW ? NSTAGE_SP * STAGE : 8 * QW * OST_ROW;
constexpr float P_OFF = 3.0f;  // P = exp2(score - rowmax + 3) <= 8 when the offset is the row maximum: 5.8 octaves of headroom to 464

__global__ __launch_bounds__(512, 2) void attn_fwd_mxfp8_sp_kernel(const unsigned char* __restrict__ Q8_, const unsigned char* __restrict__ SQ_,
                                                                  const unsigned char* __restrict__ K8_, const unsigned char* __restrict__ SK_,
                                                                  const unsigned char* __restrict__ V8T_, const unsigned char* __restrict__ SV_,
                                                                  bf16* __restrict__ O_, int Nq, int Nkv, int npad, int H, int ldq8,
                                                                  int ldk8, int ldo, int nqb, int batch, unsigned char* __restrict__ O8_,
                                                                  unsigned char* __restrict__ S8_, int ldo8, const bf16* Oadd_, int ldadd) {
  // (Oadd_: bf16 rows [batch Nq][ldadd] ADDED to this attention's bf16-rounded result before it is stored / quantised - the second
  // segment of the cross-attention, out = bf16(SDPA_text) + bf16(SDPA_image), transformer_chronoedit.py:96-107; may alias O_)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int D32 = (H * HD) >> 5;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
#pragma clang loop unroll(disable)
  for (int item = blockIdx.x; item < nqb * H * batch; item += gridDim.x) {
  const unsigned char *Q8 = Q8_, *SQ = SQ_, *K8 = K8_, *SK = SK_, *V8T = V8T_, *SV = SV_;
  bf16* O = O_;
  const bf16* Oadd = Oadd_;
  // Work order (batch folded into the item index, as in the bf16 kernel): every XCD takes its heads' FULL 256-row query blocks first,
  // sample by sample, and the remainder blocks (Nq % 256 rows: only their first waves have rows, the others merely stage) last -
  // they fill the partially occupied final round of workgroups instead of heading it.  Nq = 7200, H = 40, two samples: 2320
  // workgroups = 9.06 rounds of 256 CUs would cost ten.
  int head, qb, bz;
  {
    const int nqb_full = Nq / (QW * 8);
    if ((H & 7) == 0) {
      const int xcd = item & 7, local = item >> 3, hx_n = H >> 3;
      const int full = batch * hx_n * nqb_full;
      if (local < full) {
        bz = local / (hx_n * nqb_full);
        const int r = local % (hx_n * nqb_full);
        head = xcd + 8 * (r / nqb_full);
        qb = r % nqb_full;
      } else {
        const int l2 = local - full;
        bz = l2 / hx_n;
        head = xcd + 8 * (l2 % hx_n);
        qb = nqb_full;
      }
    } else {
      bz = item / (nqb * H);
      const int r = item % (nqb * H);
      head = r / nqb;
      qb = r % nqb;
    }
  }
  {
    const size_t b = bz;
    Q8 += b * Nq * ldq8;
    SQ += b * Nq * D32;
    K8 += b * Nkv * ldk8;
    SK += b * Nkv * D32;
    V8T += b * H * HD * npad;
    SV += b * H * HD * (npad >> 5);
    O += b * Nq * ldo;
    if (Oadd != nullptr) Oadd += b * Nq * ldadd;
  }
  const int q0 = qb * (QW * 8) + wave * QW;
  const bool active = q0 < Nq;  // wave-uniform: a wave past the last query row only stages tiles and keeps the barriers
  const int hoff = head * HD;
  const int ntiles = (Nkv + KVB - 1) / KVB;

  i32x8 qf[2];
  int sqv[2];
  {
    const int qr = min(q0 + l31, Nq - 1);
    const unsigned char* qrow = Q8 + (size_t)qr * ldq8 + hoff + 16 * hh;
    const uint32_t sw = *reinterpret_cast<const uint32_t*>(SQ + (size_t)qr * D32 + head * 4);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const u32x4 a = *reinterpret_cast<const u32x4*>(qrow + 64 * ks), b = *reinterpret_cast<const u32x4*>(qrow + 64 * ks + 32);
      qf[ks] = i32x8{(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
      sqv[ks] = (int)((sw >> (16 * ks + 8 * hh)) & 0xffu);
    }
  }
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) asm volatile("" : "+v"(qf[ks]), "+v"(sqv[ks]));  // retire the Q loads before the loop

  const int k_row = 8 * wave + (lane >> 3);
  const int k_chunk = (lane & 7) ^ ((k_row >> 1) & 7);
  const int v_row = 16 * wave + (lane >> 2);
  const int v_chunk = (lane & 3) ^ ((v_row >> 2) & 3);
  const unsigned char* vsrc = V8T + ((size_t)head * HD + v_row) * npad + v_chunk * 16;
  const unsigned char* svsrc = SV + (size_t)head * (npad >> 6) * 256 + 32 * wave + 4 * (lane & 7);
  auto stage_tile = [&](int t, int slot) __attribute__((always_inline)) {
    unsigned char* st = smem + slot * STAGE;
    const int kr = min(t * KVB + k_row, Nkv - 1);
    __builtin_amdgcn_global_load_lds((gbl_void*)(K8 + (size_t)kr * ldk8 + hoff + k_chunk * 16), (lds_void*)(st + ST_K + wave * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gbl_void*)(vsrc + (size_t)t * KVB), (lds_void*)(st + ST_V + wave * 1024), 16, 0, 0);
    if (lane < 8) {
      const int sr = min(t * KVB + 8 * wave + lane, Nkv - 1);
      __builtin_amdgcn_global_load_lds((gbl_void*)(SK + (size_t)sr * D32 + head * 4), (lds_void*)(st + ST_SK + wave * 32), 4, 0, 0);
      __builtin_amdgcn_global_load_lds((gbl_void*)(svsrc + (size_t)t * 256), (lds_void*)(st + ST_SV + wave * 32), 4, 0, 0);
    }
  };

  f32x16 oacc[4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[m][r] = 0.f;
  // The offset mc of a row is an INTEGER (exp2 domain): P = exp2(score) / 2^mc is then one v_exp_f32 and the scaled conversion
  // v_cvt_scalef32_pk_fp8_f32 (divides by the power of two of its scale operand: tools/probes/fp8_cvt_scale_probe.hip) - no subtract,
  // no accumulator-initialisation registers, and every rescale of O / l is an exact power of two.
  float l_run = 0.f, mc = 0.f, pscale = 1.0f;  // pscale = 2^mc
  i32x8 pf = {0, 0, 0, 0, 0, 0, 0, 0};         // P(t-1) as e4m3 bytes in k-slot order

  // per-lane LDS read offsets inside a stage (all per-tile addresses are these + the stage base + immediates)
  const int k_sw = (l31 >> 1) & 7;   // rows 32 f + l31: (row >> 1) & 7 does not depend on f
  const int k_off = ST_K + l31 * 128, sk_off = ST_SK + l31 * 4;
  const int v_sw = (l31 >> 2) & 3;   // rows 32 m + l31
  const int v_off = ST_V + l31 * 64 + (((2 * hh) ^ v_sw) << 4), v_off2 = ST_V + l31 * 64 + (((2 * hh + 1) ^ v_sw) << 4);
  const int sv_off = ST_SV + l31 * 2;

  // fragment fetches (two ds_read_b128 + the block-scale byte each); K rows / V rows of all fragments share one per-lane base
  auto load_k = [&](const unsigned char* st, int i, i32x8& frag, int& sc) __attribute__((always_inline)) {  // fragment i: key half f = i >> 1, k-step ks = i & 1
    const int f = i >> 1, ks = i & 1;
    const unsigned char* krow = st + k_off + f * 32 * 128;
    const int c0 = 4 * ks + hh;
    const u32x4 a = *reinterpret_cast<const u32x4*>(krow + ((c0 ^ k_sw) << 4)), b = *reinterpret_cast<const u32x4*>(krow + (((c0 + 2) ^ k_sw) << 4));
    frag = i32x8{(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
    sc = (int)((*reinterpret_cast<const uint32_t*>(st + sk_off + f * 32 * 4) >> (16 * ks + 8 * hh)) & 0xffu);
  };
  auto load_v = [&](const unsigned char* st, int m, i32x8& frag, int& sc) __attribute__((always_inline)) {
    const unsigned char* vb = st + m * 32 * 64;
    const u32x4 a = *reinterpret_cast<const u32x4*>(vb + v_off), b = *reinterpret_cast<const u32x4*>(vb + v_off2);
    frag = i32x8{(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
    sc = (int)((*reinterpret_cast<const uint16_t*>(st + sv_off + m * 64) >> (8 * hh)) & 0xffu);
  };

  // LDS-DMA runs THREE tiles ahead (five stages): at the barrier of iteration t tile t+1 is already complete, so its K fragments
  // can be fetched under the P.V MFMAs of iteration t and the S MFMAs of iteration t+1 start straight after the barrier
  // (with fragments fetched right in front of their MFMA every one of the eight MFMAs of a tile waited a full LDS round trip:
  // ~2800 cycles per tile and SIMD for 1024 cycles of matrix work).
  stage_tile(0, 0);
  stage_tile(min(1, ntiles - 1), 1);
  stage_tile(min(2, ntiles - 1), 2);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  i32x8 kf[4], vf[4];
  int ksc[4], vsc[4];
  int unit_scale = 0x7f;  // E8M0 2^0 for the P operand
  asm volatile("" : "+v"(unit_scale));
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    load_k(smem, i, kf[i], ksc[i]);
    vf[i] = i32x8{0, 0, 0, 0, 0, 0, 0, 0};
    vsc[i] = 0x7f;
  }
  // Row sums on the matrix pipe: a ones operand against P (e4m3) gives every lane its full row sum of the ROUNDED P - and a NaN
  // if any element of the tile overflowed e4m3 (the conversion returns NaN above 464: tools/probes/fp8_cvt_probe.hip), which is the
  // speculation check: one MFMA + one compare per tile instead of 32 adds per lane beside MFMAs that cannot hide them.
  i32x8 ones;
#pragma unroll
  for (int w = 0; w < 8; ++w) ones[w] = 0x38383838;  // e4m3 1.0
  asm volatile("" : "+v"(ones));
  f32x16 lsum;  // row sums of P(t-1), all 16 registers equal
#pragma unroll
  for (int r = 0; r < 16; ++r) lsum[r] = 0.f;

  auto mask_tail = [&](f32x16 (&sacc)[2], int t) __attribute__((always_inline)) {
    if ((t + 1) * KVB > Nkv) {
      const int base = t * KVB + 4 * hh;
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (base + 32 * f + (r & 3) + 8 * (r >> 2) >= Nkv) sacc[f][r] = NEG_BIG;
    }
  };
  // exact tile maximum of the row (lane ^ 32 holds the other half of the keys): raise the offset so that P <= 2^P_OFF; returns the
  // (power-of-two) factor that brings quantities at the old offset to the new one
  auto rebase = [&](const f32x16 (&sacc)[2], bool first) __attribute__((always_inline)) -> float {
    float mx = sacc[0][0];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[f][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float want = ceilf(mx - P_OFF);
    const float mc_new = first ? want : fmaxf(mc, want);  // the offset only ever rises after the first tile
    const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(mc - mc_new);
    mc = mc_new;
    pscale = __builtin_amdgcn_exp2f(mc);
    return alpha;
  };
  auto softmax_part = [&](const f32x16 (&sacc)[2], int m, i32x8& dst) __attribute__((always_inline)) {  // elements 8 m .. 8 m + 7 of the lane's 32 scores
    float p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = __builtin_amdgcn_exp2f(sacc[m >> 1][8 * (m & 1) + i]);
    s16x2 w0 = {0, 0}, w1 = {0, 0};
    w0 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(w0, p[0], p[1], pscale, false);
    w0 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(w0, p[2], p[3], pscale, true);
    w1 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(w1, p[4], p[5], pscale, false);
    w1 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(w1, p[6], p[7], pscale, true);
    dst[2 * m] = __builtin_bit_cast(int, w0);
    dst[2 * m + 1] = __builtin_bit_cast(int, w1);
  };
  // repair of tile tt (its K tile is still in LDS): S again, true row maximum, offset raised, O and l brought to the new offset,
  // P (exp2 of the score minus the offset: no fp32 overflow whatever the score) and its row sums redone
  auto exact_tile = [&](int tt) __attribute__((always_inline)) {
    const unsigned char* st = smem + (tt % NSTAGE_SP) * STAGE;
    CE_DIAG_COUNT_EXACT(g_mx_exact_hits);
    f32x16 sacc[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = i >> 1, ks = i & 1;
      i32x8 kt;
      int sc;
      load_k(st, i, kt, sc);
      f32x16 zero16;
#pragma unroll
      for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
      sacc[f] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kt, qf[ks], ks == 0 ? zero16 : sacc[f], 0, 0, 0, sc, 0, sqv[ks]);
    }
    mask_tail(sacc, tt);
    const float alpha = rebase(sacc, false);
    l_run *= alpha;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[m][r] *= alpha;
    float sum = 0.f;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      float p[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) p[i] = __builtin_amdgcn_exp2f(sacc[m >> 1][8 * (m & 1) + i] - mc);
      int w0 = 0, w1 = 0;
      w0 = __builtin_amdgcn_cvt_pk_fp8_f32(p[0], p[1], w0, false);
      w0 = __builtin_amdgcn_cvt_pk_fp8_f32(p[2], p[3], w0, true);
      w1 = __builtin_amdgcn_cvt_pk_fp8_f32(p[4], p[5], w1, false);
      w1 = __builtin_amdgcn_cvt_pk_fp8_f32(p[6], p[7], w1, true);
      pf[2 * m] = w0;
      pf[2 * m + 1] = w1;
    }
    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
    lsum = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ones, pf, zero16, 0, 0, 0, unit_scale, 0, unit_scale);
    (void)sum;
  };

  auto tile_top = [&](int t) __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // this wave's pieces of tile t+1 have landed (tile t+2's four stay in flight)
    __builtin_amdgcn_s_barrier();                     // ... everybody's have; everybody is done with tile t-2's stage
    stage_tile(min(t + 3, ntiles - 1), (t + 3) % NSTAGE_SP);
  };
  // FIRST = true (tile 0, peeled): exact offset, no P.V yet - instantiates the body without the matrix work of "tile -1", so the
  // steady-state loop carries no per-MFMA branch
  auto tile_rest = [&](int t, auto first_tag) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(first_tag)::value;
    const unsigned char* st_prev = smem + ((t + NSTAGE_SP - 1) % NSTAGE_SP) * STAGE;  // tile t-1
    const unsigned char* st_next = smem + ((t + 1) % NSTAGE_SP) * STAGE;              // tile t+1
    if (!FIRST) l_run += lsum[0];
    // ---- S^T(t) = K(t).Q^T from the fragments fetched one iteration ago.  In the MFMA shadows: the V^T(t-1) fragments of the
    // second phase.
    f32x16 sacc[2];
    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = i >> 1, ks = i & 1;
      sacc[f] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kf[i], qf[ks], ks == 0 ? zero16 : sacc[f], 0, 0, 0, ksc[i], 0, sqv[ks]);
      if (!FIRST) load_v(st_prev, i, vf[i], vsc[i]);
      __builtin_amdgcn_sched_barrier(0);
    }
    mask_tail(sacc, t);
    if (FIRST) rebase(sacc, true);
    // ---- O^T += V^T(t-1).P^T(t-1) on the matrix pipe; in the shadows P(t) = exp2(S(t)) / 2^mc -> e4m3 and the K(t+1) fragments
    i32x8 pfn;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (!FIRST) mfma_scale_acc_pinned(oacc[m], vf[m], pf, vsc[m], unit_scale);
      load_k(st_next, m, kf[m], ksc[m]);
      softmax_part(sacc, m, pfn);
      __builtin_amdgcn_sched_barrier(0);
    }
    // row sums of P(t) (rounded) on the matrix pipe, checked at the top of the next iteration
    asm volatile("s_nop 1\n\tv_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, 0, %3, %3 op_sel_hi:[0,0,0]" : "=&v"(lsum) : "v"(ones), "v"(pfn), "v"(unit_scale));
    pf = pfn;
  };
  if (active) {
    tile_top(0);
    tile_rest(0, std::true_type{});
    // The steady-state loop leaves through `break` when the speculation check of tile t-1 fails (NaN row sums = an element of P(t-1)
    // passed the e4m3 range); the repair runs OUTSIDE it and re-enters without repeating the iteration's barrier / DMA issue: as a
    // branch inside the loop it joined the common path in phi nodes that hipcc resolved with 26 register moves (and a spill reload
    // behind a vmcnt(0)) per tile.
    {
      int t = 1;
      bool skip_top = false;
      for (;;) {
        bool bad = false;
        for (; t < ntiles; ++t) {
          if (!skip_top) tile_top(t);
          skip_top = false;
          asm volatile("s_nop 7" : "+v"(lsum));  // result of the asm MFMA of the previous iteration (issued a barrier ago); no padding is inserted for asm
          if (__builtin_expect(__any(lsum[0] != lsum[0]), 0)) {
            bad = true;
            break;
          }
          tile_rest(t, std::false_type{});
        }
        if (!bad) break;
        exact_tile(t - 1);
        skip_top = true;
      }
    }
    // ---- drain: check and add the row sums of the last tile, then P(ntiles-1).V(ntiles-1) (its stage became visible at the last
    // barrier and nothing was staged over it)
    asm volatile("s_nop 15\n\ts_nop 3" : "+v"(lsum));  // the asm MFMA's result: 18 wait states before a VALU may read it (cdna guide 5.7)
    if (__builtin_expect(__any(lsum[0] != lsum[0]), 0)) exact_tile(ntiles - 1);
    l_run += lsum[0];
    {
      const unsigned char* st_last = smem + ((ntiles - 1) % NSTAGE_SP) * STAGE;
  #pragma unroll
      for (int m = 0; m < 4; ++m) {
        i32x8 vfl;
        int scl;
        load_v(st_last, m, vfl, scl);
        oacc[m] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vfl, pf, oacc[m], 0, 0, 0, scl, 0, unit_scale);
      }
    }
  } else {
    for (int t = 0; t < ntiles; ++t) tile_top(t);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  const float inv = 1.0f / l_run;  // the matrix pipe summed over all 64 k-slots: every lane holds its row's full sum
  unsigned char* ost = smem + (size_t)(wave * QW + l31) * OST_ROW;
  if (Oadd != nullptr) {
    // Second segment of a two-segment attention: this segment's result, rounded to bf16, staged row-contiguous; every lane then takes
    // 16-byte chunks (8 channels of one query row), adds the first segment's bf16 result (one coalesced 16-byte load) and either stores
    // the bf16 sum or quantises it as ce_quant_rows_mxfp8 would (a 32-channel block = 4 consecutive chunks = 4 adjacent lanes).
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const u32x2 val = {pack_bf16(oacc[m][4 * a + 0] * inv, oacc[m][4 * a + 1] * inv), pack_bf16(oacc[m][4 * a + 2] * inv, oacc[m][4 * a + 3] * inv)};
        *reinterpret_cast<u32x2*>(ost + (32 * m + 8 * a + 4 * hh) * 2) = val;
      }
    __syncthreads();
    const int ktiles = (H * HD) >> 7;
    unsigned char* O8 = O8_ != nullptr ? O8_ + (size_t)bz * Nq * ldo8 : nullptr;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = lane + 64 * i;
      const int rl = c >> 4, cc = c & 15;
      const int q = min(q0 + rl, Nq - 1);
      const bool on = q0 + rl < Nq;
      const u32x4 v = *reinterpret_cast<const u32x4*>(smem + (size_t)(wave * QW + rl) * OST_ROW + cc * 16);
      const u32x4 pv = *reinterpret_cast<const u32x4*>(Oadd + (size_t)q * ldadd + hoff + cc * 8);
      u32x4 sm;
#pragma unroll
      for (int w = 0; w < 4; ++w) sm[w] = pack_bf16(bf16lo(v[w]) + bf16lo(pv[w]), bf16hi(v[w]) + bf16hi(pv[w]));
      if (O8 == nullptr) {
        if (on) *reinterpret_cast<u32x4*>(O + (size_t)q * ldo + hoff + cc * 8) = sm;
      } else {
        float amax = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) amax = fmaxf(amax, fmaxf(fabsf(bf16lo(sm[w])), fabsf(bf16hi(sm[w]))));
        amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
        amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
        const int byte = mx_scale_byte_nosat(amax);
        const float is = mx_inv_scale(byte);
        int w0 = 0, w1 = 0;
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(bf16lo(sm[0]) * is), clamp448(bf16hi(sm[0]) * is), w0, false);
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(bf16lo(sm[1]) * is), clamp448(bf16hi(sm[1]) * is), w0, true);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(bf16lo(sm[2]) * is), clamp448(bf16hi(sm[2]) * is), w1, false);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(bf16lo(sm[3]) * is), clamp448(bf16hi(sm[3]) * is), w1, true);
        if (on) {
          *reinterpret_cast<u32x2*>(O8 + (size_t)q * ldo8 + hoff + cc * 8) = u32x2{(uint32_t)w0, (uint32_t)w1};
          if ((cc & 3) == 0) S8_[mx_gemm_scale_offset(bz * Nq + q, head * 4 + (cc >> 2), ktiles)] = (unsigned char)byte;
        }
      }
    }
  } else if (O8_ != nullptr) {
    // The out-projection's MX operand straight from the accumulators (ce_attention_mxfp8_quant): the bf16 value the plain form stores,
    // quantised as ce_quant_rows_mxfp8 would - a 32-channel block m of a query row is this lane's 16 values and the 16 of lane ^ 32.
    const int ktiles = (H * HD) >> 7;
    const int grow = bz * Nq + q0 + l31;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      uint32_t pk[8];
      float amax = 0.f;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        pk[2 * a] = pack_bf16(oacc[m][4 * a + 0] * inv, oacc[m][4 * a + 1] * inv);
        pk[2 * a + 1] = pack_bf16(oacc[m][4 * a + 2] * inv, oacc[m][4 * a + 3] * inv);
        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(bf16lo(pk[2 * a])), fabsf(bf16hi(pk[2 * a]))), fmaxf(fabsf(bf16lo(pk[2 * a + 1])), fabsf(bf16hi(pk[2 * a + 1])))));
      }
      amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
      const int byte = mx_scale_byte_nosat(amax);
      const float is = mx_inv_scale(byte);
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        int w = 0;
        w = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(bf16lo(pk[2 * a]) * is), clamp448(bf16hi(pk[2 * a]) * is), w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(bf16lo(pk[2 * a + 1]) * is), clamp448(bf16hi(pk[2 * a + 1]) * is), w, true);
        *reinterpret_cast<uint32_t*>(ost + 32 * m + 8 * a + 4 * hh) = (uint32_t)w;
      }
      if (hh == 0 && q0 + l31 < Nq) S8_[mx_gemm_scale_offset(grow, head * 4 + m, ktiles)] = (unsigned char)byte;
    }
    __syncthreads();
    unsigned char* O8 = O8_ + (size_t)bz * Nq * ldo8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = lane + 64 * i;
      const int rl = c >> 3, cc = c & 7;
      const int q = min(q0 + rl, Nq - 1);
      const u32x4 v = *reinterpret_cast<const u32x4*>(smem + (size_t)(wave * QW + rl) * OST_ROW + cc * 16);
      if (q0 + rl < Nq) *reinterpret_cast<u32x4*>(O8 + (size_t)q * ldo8 + hoff + cc * 16) = v;
    }
  } else {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const u32x2 val = {pack_bf16(oacc[m][4 * a + 0] * inv, oacc[m][4 * a + 1] * inv), pack_bf16(oacc[m][4 * a + 2] * inv, oacc[m][4 * a + 3] * inv)};
        *reinterpret_cast<u32x2*>(ost + (32 * m + 8 * a + 4 * hh) * 2) = val;
      }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = lane + 64 * i;
      const int rl = c >> 4, cc = c & 15;
      const int q = min(q0 + rl, Nq - 1);  // clamped address, predicated store: no per-chunk branch around the LDS read
      const u32x4 v = *reinterpret_cast<const u32x4*>(smem + (size_t)(wave * QW + rl) * OST_ROW + cc * 16);
      if (q0 + rl < Nq) *reinterpret_cast<u32x4*>(O + (size_t)q * ldo + hoff + cc * 8) = v;
    }
  }
  __syncthreads();
  }  // item
}

}  // namespace

CE_API int ce_rmsnorm_rope_mxfp8(const void* x, const float* w, const float* cos_sin, void* q8, void* scale8, int M, int D, int ldx,
                                     int ldq, int head_dim, float eps, int rope_rows, float post_scale, hipStream_t stream) {
  if (!x || !w || !q8 || !scale8) return CE_ERR_ARG;
  if (M <= 0 || D <= 0 || (D & 31) || D > 64 * 8 * ROW_MAXC || (ldx & 7) || (ldq & 7) || (head_dim & 31) || D % head_dim) return CE_ERR_SHAPE;
  if (D == 64 * 8 * ROW_MAXC)
    hipLaunchKernelGGL(rmsnorm_rope_mxfp8_kernel<true>, dim3((M + 3) / 4), dim3(256), 0, stream, (const bf16*)x, w, cos_sin, (unsigned char*)q8,
                       (unsigned char*)scale8, M, D, ldx, ldq, head_dim, eps, rope_rows, post_scale);
  else
    hipLaunchKernelGGL(rmsnorm_rope_mxfp8_kernel<false>, dim3((M + 3) / 4), dim3(256), 0, stream, (const bf16*)x, w, cos_sin, (unsigned char*)q8,
                       (unsigned char*)scale8, M, D, ldx, ldq, head_dim, eps, rope_rows, post_scale);
  return (int)hipGetLastError();
}

CE_API int ce_v_mxfp8_transpose(const void* v, int ldv, void* v8t, void* sv, int n_tokens, int batch, int H, int npad, hipStream_t stream) {
  if (!v || !v8t || !sv) return CE_ERR_ARG;
  if (n_tokens <= 0 || batch <= 0 || H <= 0 || (ldv & 7) || (npad & 63) || npad < n_tokens) return CE_ERR_SHAPE;
  hipLaunchKernelGGL(v_mxfp8_transpose_kernel, dim3(npad / KVB, H, batch), dim3(256), 0, stream, (const bf16*)v, ldv, (unsigned char*)v8t,
                     (unsigned char*)sv, n_tokens, H, npad);
  return (int)hipGetLastError();
}

#ifdef CE_DIAGNOSTICS
// (diagnostic build) exact-route count of the MXFP8 kernel since the last reset; called by ce_diag_attention_exact_route_hits (ce_attn.hip)
extern "C" int ce_diag_mx_exact_hits(unsigned long long* out, int reset) {
  unsigned long long v = 0ull, z = 0ull;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_mx_exact_hits), sizeof(v)) != hipSuccess) return CE_ERR_ARG;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_mx_exact_hits), &z, sizeof(z)) != hipSuccess) return CE_ERR_ARG;
  *out = v;
  return CE_OK;
}
#endif
CE_KNOB g_mxfp8_persist = 512;  // workgroups of the persistent form (0: one workgroup per work item); a multiple of 8
CE_KNOB g_mxfp8_variant = 1;    // 0: plain kernel (exact running maximum every tile), 1: software-pipelined (default)
#ifdef CE_DIAGNOSTICS
CE_API int ce_set_attention_mxfp8_persistent(int n) {
  const int old = g_mxfp8_persist;
  if (n >= 0 && (n & 7) == 0) g_mxfp8_persist = n;
  return old;
}
CE_API int ce_set_attention_mxfp8_variant(int v) {
  const int old = g_mxfp8_variant;
  if (v == 0 || v == 1) g_mxfp8_variant = v;
  return old;
}
#endif

static int attention_mxfp8_launch(const void* q8, const void* sq, const void* k8, const void* sk, const void* v8t, const void* sv, void* O,
                                  int Nq, int Nkv, int npad, int H, int head_dim, int ldq8, int ldk8, int ldo, int batch, void* O8, void* S8,
                                  int ldo8, hipStream_t stream, const void* o_add = nullptr, int ldadd = 0) {
  if (o_add && (ldadd & 7)) return CE_ERR_ALIGN;
  if (!q8 || !sq || !k8 || !sk || !v8t || !sv || (!O && !O8) || (O8 && !S8)) return CE_ERR_ARG;
  if (head_dim != HD || Nq <= 0 || Nkv <= 0 || H <= 0 || batch <= 0 || (npad & 63) || npad < Nkv || npad - Nkv >= KVB) return CE_ERR_SHAPE;
  if ((ldq8 & 15) || (ldk8 & 15) || (O && (ldo & 7)) || (O8 && (ldo8 & 15))) return CE_ERR_ALIGN;
  const int nqb = (Nq + QW * 8 - 1) / (QW * 8);
  static bool attr_[CE_MAX_DEVICES] = {};
  bool& attr = attr_[ce_device_slot()];
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)attn_fwd_mxfp8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    (void)hipFuncSetAttribute((const void*)attn_fwd_mxfp8_sp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_SP);
    attr = true;
  }
  if (g_mxfp8_variant == 0 && !O8 && !o_add)
    hipLaunchKernelGGL(attn_fwd_mxfp8_kernel, dim3(H * nqb, batch), dim3(512), SMEM, stream, (const unsigned char*)q8, (const unsigned char*)sq,
                       (const unsigned char*)k8, (const unsigned char*)sk, (const unsigned char*)v8t, (const unsigned char*)sv, (bf16*)O, Nq, Nkv,
                       npad, H, ldq8, ldk8, ldo, nqb);
  else
    hipLaunchKernelGGL(attn_fwd_mxfp8_sp_kernel, dim3(g_mxfp8_persist > 0 && H * nqb * batch > g_mxfp8_persist ? g_mxfp8_persist : H * nqb * batch), dim3(512), SMEM_SP, stream, (const unsigned char*)q8,
                       (const unsigned char*)sq, (const unsigned char*)k8, (const unsigned char*)sk, (const unsigned char*)v8t,
                       (const unsigned char*)sv, (bf16*)O, Nq, Nkv, npad, H, ldq8, ldk8, ldo, nqb, batch, (unsigned char*)O8, (unsigned char*)S8,
                       ldo8, (const bf16*)o_add, ldadd);
  return (int)hipGetLastError();
}

// Second segment of a two-segment attention (the cross-attention's image segment, transformer_chronoedit.py:96-107): as ce_attention_mxfp8 /
// ce_attention_mxfp8_quant, with the bf16 rows o_add [batch Nq][ldadd] - the first segment's result - added to this segment's bf16-rounded
// result; the sum is stored as bf16 (O; may be o_add itself) or as the out-projection's MX operand (o8 + scale8; exactly one of the two).
CE_API int ce_attention_mxfp8_add(const void* q8, const void* sq, const void* k8, const void* sk, const void* v8t, const void* sv,
                                      const void* o_add, int ldadd, void* O, int ldo, void* o8, void* scale8, int ldo8, int Nq, int Nkv, int npad,
                                      int H, int head_dim, int ldq8, int ldk8, int batch, hipStream_t stream) {
  if (!o_add || (O == nullptr) == (o8 == nullptr)) return CE_ERR_ARG;
  return attention_mxfp8_launch(q8, sq, k8, sk, v8t, sv, O, Nq, Nkv, npad, H, head_dim, ldq8, ldk8, ldo, batch, o8, scale8, ldo8, stream, o_add, ldadd);
}

CE_API int ce_attention_mxfp8(const void* q8, const void* sq, const void* k8, const void* sk, const void* v8t, const void* sv, void* O,
                                  int Nq, int Nkv, int npad, int H, int head_dim, int ldq8, int ldk8, int ldo, int batch, hipStream_t stream) {
  if (!O) return CE_ERR_ARG;
  return attention_mxfp8_launch(q8, sq, k8, sk, v8t, sv, O, Nq, Nkv, npad, H, head_dim, ldq8, ldk8, ldo, batch, nullptr, nullptr, 0, stream);
}

// The same attention with its output written as the out-projection's MX operand: o8 e4m3 [batch Nq][ldo8] + E8M0 block scales in the
// tiled layout of ce_gemm_mxfp8 (rows = batch Nq, K = H head_dim) - bit-identical to ce_attention_mxfp8 followed by ce_quant_rows_mxfp8
// (the software-pipelined kernel whatever ce_set_attention_mxfp8_variant says).
CE_API int ce_attention_mxfp8_quant(const void* q8, const void* sq, const void* k8, const void* sk, const void* v8t, const void* sv,
                                        void* o8, void* scale8, int Nq, int Nkv, int npad, int H, int head_dim, int ldq8, int ldk8, int ldo8,
                                        int batch, hipStream_t stream) {
  if (!o8 || !scale8) return CE_ERR_ARG;
  return attention_mxfp8_launch(q8, sq, k8, sk, v8t, sv, nullptr, Nq, Nkv, npad, H, head_dim, ldq8, ldk8, 0, batch, o8, scale8, ldo8, stream);
}
