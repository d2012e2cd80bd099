// 256x256x128 fp8 (OCP e4m3) GEMM with per-row / per-column scales, and the row quantiser that feeds it — BASELINE.json
// configs[4] ("fp8 weights + activations on the CDNA4 fp8 matrix path"; the reference itself has no fp8 inference path, so
// the arithmetic contract is defined here and pinned by tests/test_fp8_gpu.py against fp32 math on the dequantised operands):
//     C[m][n] = epilogue( sa[m] * sw[n] * sum_k Aq[m][k] * Wq[n][k] + bias[n] )      Aq, Wq fp8 e4m3, sa / sw fp32, C bf16
//     ce_quant_rows_fp8:  s[m] = max_k |x[m][k]| / 448 (1 if the row is zero),  q[m][k] = fp8_rne(x[m][k] / s[m])
// Same structure as gemm_bf16_256 (ce_gemm256.hip: 8 waves as 2 x 4, eight 16-KiB half-tile LDS slots filled by LDS-DMA with
// the source-side chunk swizzle, 4 phases per K-tile with counted vmcnt): a 128-byte LDS row is now 128 fp8 = ONE k-step of
// v_mfma_scale_f32_16x16x128_f8f6f4 (block scales fixed at 2^0; probe: tools/probes/mx_probe.hip).  Per byte moved the
// kernel does twice the flops of the bf16 one, and the MX instruction runs at twice the bf16 MFMA rate.
// Operand packing: lane (r = lane & 15, g = lane >> 4) feeds row r with the 32 bytes k = 32 g + [0, 32) of the k-step - the
// two 16-B chunks 2g and 2g + 1 of the LDS row.  Any packing that is the same for A and W is correct (sum over k).
#include "ce_common.h"
#include "ce_gemm_epi.h"

namespace {

constexpr int BM = 256, BN = 256, BKB = 128;  // K-tile in BYTES (= fp8 elements)
constexpr int SLOT = 128 * BKB;               // 16 KiB half-tile
constexpr int LDS_TILES = 8 * SLOT;
constexpr int CROW = BN * 2 + 16;
constexpr int LDS_BYTES = LDS_TILES > 128 * CROW ? LDS_TILES : 128 * CROW;
constexpr int S_A0 = 0, S_A1 = 1, S_B0 = 2, S_B1 = 3;

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;
typedef __attribute__((ext_vector_type(8))) int i32x8;

__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

struct Stager {
  uint32_t a_off[2][2], w_off[2][2];
  const char* a_base;
  const char* w_base;
  int wave;
  int kt_last;
};

template <int SLOT_ID>
__device__ __forceinline__ void stage_half(unsigned char* smem, const Stager& s, int tile) {
  constexpr int half = SLOT_ID & 1;
  constexpr bool isB = (SLOT_ID & 2) != 0;
  const int t = tile < s.kt_last ? tile : s.kt_last;
  const char* base = (isB ? s.w_base : s.a_base) + (size_t)t * BKB;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const uint32_t off = isB ? s.w_off[half][r] : s.a_off[half][r];
    unsigned char* dst = smem + SLOT_ID * SLOT + (r * 8 + s.wave) * 1024;
    __builtin_amdgcn_global_load_lds((gbl_void*)(base + off), (lds_void*)dst, 16, 0, 0);
  }
}

// half H (0 / 1) of the 32-byte operand of every fragment: chunk 2 fg + H of the row
template <int SLOT_ID, int H>
__device__ __forceinline__ void read_a(const unsigned char* smem, int wm, int fr, int fg, u32x4 (&a)[4]) {
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const int row = wm * 64 + f * 16 + fr;
    a[f] = *reinterpret_cast<const u32x4*>(smem + SLOT_ID * SLOT + row * BKB + (((2 * fg + H) ^ swz(row)) << 4));
  }
}
template <int SLOT_ID, int H>
__device__ __forceinline__ void read_b(const unsigned char* smem, int wn, int fr, int fg, u32x4 (&b)[2]) {
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int row = wn * 32 + g * 16 + fr;
    b[g] = *reinterpret_cast<const u32x4*>(smem + SLOT_ID * SLOT + row * BKB + (((2 * fg + H) ^ swz(row)) << 4));
  }
}

// 8 MX MFMAs: the whole k-step (128 deep) of a 64x32 quadrant
__device__ __forceinline__ void mma_quad(f32x4 (&acc)[4][2], const u32x4 (&a0)[4], const u32x4 (&a1)[4], const u32x4 (&b0)[2],
                                         const u32x4 (&b1)[2]) {
  __builtin_amdgcn_s_setprio(1);
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const i32x8 av = {(int)a0[f][0], (int)a0[f][1], (int)a0[f][2], (int)a0[f][3], (int)a1[f][0], (int)a1[f][1], (int)a1[f][2], (int)a1[f][3]};
      const i32x8 bv = {(int)b0[g][0], (int)b0[g][1], (int)b0[g][2], (int)b0[g][3], (int)b1[g][0], (int)b1[g][1], (int)b1[g][2], (int)b1[g][3]};
      // W fragment first (as in the bf16 kernel): the lane then owns FOUR CONSECUTIVE COLUMNS of one row of C - 8-byte LDS stores and
      // 16-byte scale / bias loads in the epilogue instead of 2-byte stores (the A-first form left 128 ds_write_b16 per thread and tile)
      acc[f][g] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bv, av, acc[f][g], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
  __builtin_amdgcn_s_setprio(0);
}

#define F8_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0)
#define F8_VM(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#define F8_BAR() __builtin_amdgcn_s_barrier()

template <int EPI>
__global__ __launch_bounds__(512) void gemm_fp8_256(const unsigned char* __restrict__ A, const unsigned char* __restrict__ W,
                                                    bf16* __restrict__ C, const float* __restrict__ sa, const float* __restrict__ sw,
                                                    const float* __restrict__ bias, const float* __restrict__ gate,
                                                    const bf16* __restrict__ res, int M, int N, int K, int lda, int ldw, int ldc,
                                                    int ldres, int gate_rows, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int fr = lane & 15, fg = lane >> 4;

  const int wg = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  int m0, n0;
  {
    constexpr int GROUP = 4;
    const int group_sz = GROUP * tiles_n, gid = wg / group_sz, first_m = gid * GROUP;
    const int gm = min(tiles_m - first_m, GROUP);
    m0 = (first_m + (wg % group_sz) % gm) * BM;
    n0 = ((wg % group_sz) / gm) * BN;
  }

  Stager st;
  st.a_base = reinterpret_cast<const char*>(A);
  st.w_base = reinterpret_cast<const char*>(W);
  st.wave = wave;
  st.kt_last = K / BKB - 1;
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int row = (r * 8 + wave) * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ swz(row);
      st.a_off[h][r] = (uint32_t)min(m0 + h * 128 + row, M - 1) * (uint32_t)lda + chunk * 16;
      st.w_off[h][r] = (uint32_t)min(n0 + h * 128 + row, N - 1) * (uint32_t)ldw + chunk * 16;
    }

  f32x4 acc[2][2][4][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int g = 0; g < 2; ++g) acc[i][j][f][g] = f32x4{0.f, 0.f, 0.f, 0.f};

  stage_half<S_A0>(smem, st, 0);
  stage_half<S_B0>(smem, st, 0);
  stage_half<S_B1>(smem, st, 0);
  stage_half<S_A1>(smem, st, 0);
  stage_half<4 + S_A0>(smem, st, 1);
  stage_half<4 + S_B0>(smem, st, 1);
  stage_half<4 + S_B1>(smem, st, 1);
  F8_VM(4);
  F8_BAR();

  u32x4 ra0[4], ra1[4], b0k0[2], b0k1[2], b1k0[2], b1k1[2];
  read_a<S_A0, 0>(smem, wm, fr, fg, ra0);
  read_a<S_A0, 1>(smem, wm, fr, fg, ra1);
  read_b<S_B0, 0>(smem, wn, fr, fg, b0k0);
  read_b<S_B0, 1>(smem, wn, fr, fg, b0k1);
  read_b<S_B1, 0>(smem, wn, fr, fg, b1k0);
  read_b<S_B1, 1>(smem, wn, fr, fg, b1k1);

#define F8_STAGE(SLOT_EVEN, SLOT_ODD, CURV, TILEV) \
  if (CURV == 0) stage_half<SLOT_EVEN>(smem, st, (TILEV)); else stage_half<SLOT_ODD>(smem, st, (TILEV));
  // slot / phase timetable of gemm_bf16_256 (fragment reads and stages in the same phases, so its hazard argument carries
  // over); a quadrant's 8 MFMAs need both halves of every operand, so the refills follow the MFMAs of the phase
#define F8_TILE_PHASES(CUR, NXT, TILE)                                   \
  /* phase 1: Q00 */                                                     \
  F8_BAR();                                                              \
  F8_LGKM0();                                                            \
  mma_quad(acc[0][0], ra0, ra1, b0k0, b0k1);                             \
  F8_STAGE(4 + S_A1, S_A1, CUR, (TILE) + 1)                              \
  /* phase 2: Q01, then A-sub1 of this tile into the A registers */      \
  F8_VM(8);                                                              \
  F8_BAR();                                                              \
  mma_quad(acc[0][1], ra0, ra1, b1k0, b1k1);                             \
  read_a<CUR * 4 + S_A1, 0>(smem, wm, fr, fg, ra0);                      \
  read_a<CUR * 4 + S_A1, 1>(smem, wm, fr, fg, ra1);                      \
  F8_STAGE(S_A0, 4 + S_A0, CUR, (TILE) + 2)                              \
  /* phase 3: Q11 */                                                     \
  F8_BAR();                                                              \
  F8_LGKM0();                                                            \
  mma_quad(acc[1][1], ra0, ra1, b1k0, b1k1);                             \
  F8_STAGE(S_B0, 4 + S_B0, CUR, (TILE) + 2)                              \
  /* phase 4: Q10; the next tile's B1 before, its A0 / B0 after */       \
  F8_VM(6);                                                              \
  F8_BAR();                                                              \
  read_b<NXT * 4 + S_B1, 0>(smem, wn, fr, fg, b1k0);                     \
  read_b<NXT * 4 + S_B1, 1>(smem, wn, fr, fg, b1k1);                     \
  mma_quad(acc[1][0], ra0, ra1, b0k0, b0k1);                             \
  read_a<NXT * 4 + S_A0, 0>(smem, wm, fr, fg, ra0);                      \
  read_a<NXT * 4 + S_A0, 1>(smem, wm, fr, fg, ra1);                      \
  read_b<NXT * 4 + S_B0, 0>(smem, wn, fr, fg, b0k0);                     \
  read_b<NXT * 4 + S_B0, 1>(smem, wn, fr, fg, b0k1);                     \
  F8_STAGE(S_B1, 4 + S_B1, CUR, (TILE) + 2)

  const int npairs = (K / BKB) >> 1;
  for (int it = 0; it < npairs; ++it) {
    const int t = 2 * it;
    F8_TILE_PHASES(0, 1, t)
    F8_TILE_PHASES(1, 0, t + 1)
  }
#undef F8_TILE_PHASES
#undef F8_STAGE
  F8_VM(0);
  F8_LGKM0();
  F8_BAR();

  // ---- epilogue: scales, bias -> bf16 -> LDS (two passes of 128 rows) -> row-contiguous activation / residual math and stores.
  // Column scales / bias of the lane's four (j, g) column groups: the same for both passes, 16-byte loads on clamped addresses,
  // issued together and without a per-lane guard (a guarded load compiles to its own branch with a vmcnt(0) behind it).
  // (the gated-residual epilogue is at its register limit: there the 32 registers are re-loaded per pass instead of living across both)
  f32x4 swv[2][2], bvv[2][2];
  auto load_cols = [&](int opaque) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        int nc = min(n0 + j * 128 + wn * 32 + g * 16 + fg * 4, N - 4) + opaque;  // n is a multiple of 4 and N of 8
        swv[j][g] = *reinterpret_cast<const f32x4*>(sw + nc);
        bvv[j][g] = bias != nullptr ? *reinterpret_cast<const f32x4*>(bias + nc) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
  };
  if (EPI != EPI_GATE_RES) load_cols(0);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (i == 1) __syncthreads();
    if (EPI == EPI_GATE_RES) {
      int zero = 0;
      asm volatile("" : "+v"(zero));  // keeps the second pass's loads from being merged with (and kept live since) the first's
      load_cols(zero);
    }
    float sav[4];  // row scale of this lane's row in each of the four 16-row fragments
#pragma unroll
    for (int f = 0; f < 4; ++f) sav[f] = sa[min(m0 + i * 128 + wm * 64 + f * 16 + fr, M - 1)];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int cl = j * 128 + wn * 32 + g * 16 + fg * 4;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          const int rl = wm * 64 + f * 16 + fr;
          const f32x4 v = acc[i][j][f][g];
          const u32x2 pk = {pack_bf16(v[0] * (sav[f] * swv[j][g][0]) + bvv[j][g][0], v[1] * (sav[f] * swv[j][g][1]) + bvv[j][g][1]),
                            pack_bf16(v[2] * (sav[f] * swv[j][g][2]) + bvv[j][g][2], v[3] * (sav[f] * swv[j][g][3]) + bvv[j][g][3])};
          *reinterpret_cast<u32x2*>(smem + rl * CROW + cl * 2) = pk;
        }
      }
    __syncthreads();
    epi_chunks<EPI, 8>(smem, CROW, [&](int tt, int& rl, int& cc, int& mr) { const int c = tid + 512 * tt; rl = mr = c >> 5; cc = c & 31; },
                       m0 + i * 128, n0, C, gate, res, M, N, ldc, ldres, gate_rows);
  }
}

// one wave per row: amax -> scale -> fp8 (v_cvt_pk_fp8_f32, OCP e4m3 on gfx950, round to nearest even)
// NCHL > 0: K == 64 * 8 * NCHL, the row sits in registers (one HBM read, all loads issued back to back - a load inside a
// run-time loop or a per-chunk guard is waited for before the next one is issued); NCHL == 0: any K, two passes over the row.
template <int NCHL>
__global__ __launch_bounds__(256) void quant_rows_fp8_kernel(const bf16* __restrict__ x, unsigned char* __restrict__ q,
                                                             float* __restrict__ scale, int M, int K, int ldx, int ldq) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nch = K >> 3;
  const bf16* xr = x + (size_t)row * ldx;
  constexpr int NR = NCHL > 0 ? NCHL : 1;
  u32x4 raw[NR];
  float amax = 0.f;
  if (NCHL > 0) {
#pragma unroll
    for (int i = 0; i < NR; ++i) raw[i] = *reinterpret_cast<const u32x4*>(xr + (lane + 64 * i) * 8);
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) amax = fmaxf(amax, fmaxf(fabsf(bf16lo(raw[i][j])), fabsf(bf16hi(raw[i][j]))));
  } else {
    for (int c = lane; c < nch; c += 64) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(xr + c * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j) amax = fmaxf(amax, fmaxf(fabsf(bf16lo(v[j])), fabsf(bf16hi(v[j]))));
    }
  }
  amax = wave_max(amax);
  const float s = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
  const float inv = 1.0f / s;
  if (lane == 0) scale[row] = s;
  unsigned char* qr = q + (size_t)row * ldq;
  auto put = [&](int c, u32x4 v) {
    int w0 = 0, w1 = 0;
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(bf16lo(v[0]) * inv, bf16hi(v[0]) * inv, w0, false);
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(bf16lo(v[1]) * inv, bf16hi(v[1]) * inv, w0, true);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(bf16lo(v[2]) * inv, bf16hi(v[2]) * inv, w1, false);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(bf16lo(v[3]) * inv, bf16hi(v[3]) * inv, w1, true);
    u32x2 o = {(uint32_t)w0, (uint32_t)w1};
    *reinterpret_cast<u32x2*>(qr + c * 8) = o;
  };
  if (NCHL > 0) {
#pragma unroll
    for (int i = 0; i < NR; ++i) put(lane + 64 * i, raw[i]);
  } else {
    for (int c = lane; c < nch; c += 64) put(c, *reinterpret_cast<const u32x4*>(xr + c * 8));  // second read: L2-resident
  }
}

// MX form: one wave per row, every 32 consecutive elements (= 4 lanes x 8) get their own E8M0 scale: no row-wide amax, one pass.
// Scales go to the TILED layout the MX GEMM reads (mx_gemm_scale_offset, ce_common.h).
// NCHL > 0: K == 64 * 8 * NCHL, every load of the row issued back to back (a load inside a run-time loop is waited for before the next
// one is issued: the generic form ran at 60 % of the per-row kernel's rate on the step's widths); NCHL == 0: any K, four chunks in flight.
template <int NCHL>
__global__ __launch_bounds__(256) void quant_rows_mxfp8_kernel(const bf16* __restrict__ x, unsigned char* __restrict__ q,
                                                               unsigned char* __restrict__ sc, int M, int K, int ldx, int ldq, int w_order) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nch = K >> 3, ktiles = K >> 7;
  const bf16* xr = x + (size_t)row * ldx;
  unsigned char* qr = q + (size_t)row * ldq;
  constexpr int NR = NCHL > 0 ? NCHL : 4;
  for (int c0 = 0; c0 < nch; c0 += 64 * NR) {
    u32x4 raw[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int c = c0 + lane + 64 * i;
      raw[i] = (NCHL > 0 || c < nch) ? *reinterpret_cast<const u32x4*>(xr + c * 8) : u32x4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int c = c0 + lane + 64 * i;
      float amax = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) amax = fmaxf(amax, fmaxf(fabsf(bf16lo(raw[i][j])), fabsf(bf16hi(raw[i][j]))));
      amax = fmaxf(amax, __shfl_xor(amax, 1, 64));  // a block = chunks 4a .. 4a+3 = lanes 4a' .. 4a'+3 (K % 32 == 0: whole blocks only)
      amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
      if (c < nch) {
        const int byte = mx_scale_byte_nosat(amax);
        const float inv = mx_inv_scale(byte);
        const u32x4 v = raw[i];
        int w0 = 0, w1 = 0;
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(bf16lo(v[0]) * inv), clamp448(bf16hi(v[0]) * inv), w0, false);
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(bf16lo(v[1]) * inv), clamp448(bf16hi(v[1]) * inv), w0, true);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(bf16lo(v[2]) * inv), clamp448(bf16hi(v[2]) * inv), w1, false);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(bf16lo(v[3]) * inv), clamp448(bf16hi(v[3]) * inv), w1, true);
        const u32x2 o = {(uint32_t)w0, (uint32_t)w1};
        *reinterpret_cast<u32x2*>(qr + c * 8) = o;
        if ((lane & 3) == 0) sc[w_order ? mx_gemm_wscale_offset(row, c >> 2, ktiles) : mx_gemm_scale_offset(row, c >> 2, ktiles)] = (unsigned char)byte;
      }
    }
  }
}

}  // namespace

/* x bf16 [M][ldx] -> q e4m3 bytes [M][ldq] + E8M0 block scales (one per 32 consecutive elements of a row, scale = 2^(floor(log2 amax) - 8),
 * elements RNE(x / scale) clamped to +-448: the OCP MX contract of oracle.dit_oracle.mx_quant) in the tiled layout of ce_gemm_mxfp8's A
 * operand (ce_quant_rows_mxfp8: activations) or of its W operand (ce_quant_rows_mxfp8_w: weights, once at load time):
 * scale8 holds ceil(M / 128) * (K / 128) * 512 bytes.  K % 128 == 0. */
static int quant_rows_mxfp8_launch(const void* x, void* q, void* scale8, int M, int K, int ldx, int ldq, int w_order, hipStream_t stream) {
  if (!x || !q || !scale8) return CE_ERR_ARG;
  if (M <= 0 || K <= 0 || (K & 127)) return CE_ERR_SHAPE;
  if ((ldx & 7) || (ldq & 7)) return CE_ERR_ALIGN;
#define MX_QUANT(NC) \
  hipLaunchKernelGGL(quant_rows_mxfp8_kernel<NC>, dim3((M + 3) / 4), dim3(256), 0, stream, (const bf16*)x, (unsigned char*)q, (unsigned char*)scale8, M, K, ldx, ldq, w_order)
  if (K == 64 * 8 * 10) MX_QUANT(10);       // 5120
  else if (K == 64 * 8 * 27) MX_QUANT(27);  // 13824
  else MX_QUANT(0);
#undef MX_QUANT
  return (int)hipGetLastError();
}
CE_API int ce_quant_rows_mxfp8(const void* x, void* q, void* scale8, int M, int K, int ldx, int ldq, hipStream_t stream) {
  return quant_rows_mxfp8_launch(x, q, scale8, M, K, ldx, ldq, 0, stream);
}
CE_API int ce_quant_rows_mxfp8_w(const void* x, void* q, void* scale8, int M, int K, int ldx, int ldq, hipStream_t stream) {
  return quant_rows_mxfp8_launch(x, q, scale8, M, K, ldx, ldq, 1, stream);
}

CE_API int ce_quant_rows_fp8(const void* x, void* q, float* scale, int M, int K, int ldx, int ldq, hipStream_t stream) {
  if (!x || !q || !scale) return CE_ERR_ARG;
  if (M <= 0 || K <= 0) return CE_ERR_SHAPE;
  if ((K & 7) || (ldx & 7) || (ldq & 7)) return CE_ERR_ALIGN;
#define F8_QUANT(NC) \
  hipLaunchKernelGGL(quant_rows_fp8_kernel<NC>, dim3((M + 3) / 4), dim3(256), 0, stream, (const bf16*)x, (unsigned char*)q, scale, M, K, ldx, ldq)
  if (K == 64 * 8 * 10) F8_QUANT(10);       // 5120
  else if (K == 64 * 8 * 27) F8_QUANT(27);  // 13824
  else F8_QUANT(0);
#undef F8_QUANT
  return (int)hipGetLastError();
}

extern "C" int ce_gemm_fp8w4_launch(const void* Aq, const void* Wq, void* C, const float* sa, const float* sw, const float* bias,
                                    int epilogue, const float* gate, const void* res, int M, int N, int K, int lda, int ldw, int ldc,
                                    int ldres, int gate_rows, hipStream_t stream);

// main loop of ce_gemm_fp8: 1 = one wave per SIMD (ce_gemm_fp8w4.hip; the default: +2 ... +8 % on the step's shapes,
// profiles/r03_gemm_fp8_variants_ab.txt), 0 = the 8-wave / 4-phase loop of this file
CE_KNOB g_fp8_variant = 1;
#ifdef CE_DIAGNOSTICS
CE_API int ce_set_gemm_fp8_variant(int v) {
  const int old = g_fp8_variant;
  if (v == 0 || v == 1) g_fp8_variant = v;
  return old;
}
#endif

CE_API int ce_gemm_fp8(const void* Aq, const void* Wq, void* C, const float* sa, const float* sw, const float* bias, int epilogue,
                           const float* gate, const void* res, int M, int N, int K, int lda, int ldw, int ldc, int ldres,
                           int gate_rows, hipStream_t stream) {
  if (!Aq || !Wq || !C || !sa || !sw) return CE_ERR_ARG;
  if (M <= 0 || N <= 0 || K <= 0 || (K % (2 * BKB)) || (N & 7)) return CE_ERR_SHAPE;
  if ((lda & 15) || (ldw & 15) || (ldc & 7)) return CE_ERR_ALIGN;
  if ((long long)M * lda >= (1ll << 32) || (long long)N * ldw >= (1ll << 32)) return CE_ERR_SHAPE;  // 32-bit DMA offsets
  if (epilogue == EPI_GATE_RES && (!res || (ldres & 7))) return CE_ERR_ARG;
  // (the one-wave-per-SIMD loop's gated-residual epilogue holds ONE or TWO samples' gate rows per tile and stores through 32-bit offsets)
  const bool w4_gate_ok = epilogue != EPI_GATE_RES || ((gate == nullptr || gate_rows == 0 || gate_rows >= BM) && (long long)M * ldc * 2 < (1ll << 32));
  if (g_fp8_variant == 1 && w4_gate_ok && (epilogue == EPI_BIAS || epilogue == EPI_BIAS_GELU || epilogue == EPI_GATE_RES))
    return ce_gemm_fp8w4_launch(Aq, Wq, C, sa, sw, bias, epilogue, gate, res, M, N, K, lda, ldw, ldc, ldres, gate_rows, stream);
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  dim3 grid(tiles_m * tiles_n), block(512);
  static bool attr_done_[CE_MAX_DEVICES][3] = {};
  bool* attr_done = attr_done_[ce_device_slot()];
#define F8_LAUNCH(E)                                                                                                        \
  do {                                                                                                                      \
    if (!attr_done[E]) {                                                                                                    \
      (void)hipFuncSetAttribute((const void*)gemm_fp8_256<E>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);       \
      attr_done[E] = true;                                                                                                  \
    }                                                                                                                       \
    hipLaunchKernelGGL((gemm_fp8_256<E>), grid, block, LDS_BYTES, stream, (const unsigned char*)Aq, (const unsigned char*)Wq, \
                       (bf16*)C, sa, sw, bias, gate, (const bf16*)res, M, N, K, lda, ldw, ldc, ldres, gate_rows, tiles_m, tiles_n); \
  } while (0)
  switch (epilogue) {
    case EPI_BIAS: F8_LAUNCH(EPI_BIAS); break;
    case EPI_BIAS_GELU: F8_LAUNCH(EPI_BIAS_GELU); break;
    case EPI_GATE_RES: F8_LAUNCH(EPI_GATE_RES); break;
    default: return CE_ERR_ARG;
  }
#undef F8_LAUNCH
  return (int)hipGetLastError();
}
