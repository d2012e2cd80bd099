// 256x256x64 bf16 GEMM, ONE WAVE PER SIMD: 4 waves (2 x 2) x up to 512 registers, wave tile 128 x 128.
// Same contract, epilogues, split-K tail and segmented operands as ce_gemm256.hip (`ce_set_gemm_variant(3 / 4)` selects it).
//
// Why a second main loop: the 8-wave kernel reads 192 KiB of LDS fragments per K-tile and CU (a 64x32 quadrant = 12
// ds_read_b128 per 32 MFMAs and k-step) and runs four workgroup barriers per K-tile between waves that share SIMDs pairwise.
// Here a wave owns a 128x128 quadrant - 16 fragment reads per 64 MFMAs and k-step, 128 KiB per K-tile and CU (-33 %) - its
// 256 accumulator registers live in the AGPR half of the register file and the VGPR half holds THREE fragment sets, so that a
// whole K-tile's fragments are fetched one unit ahead; two barriers per K-tile, no intra-SIMD arbitration at all.
//
//  * LDS: A ring of NSA (3 or 2) K-tile stages [256 rows][128 B] + W ring of 2 stages = 160 KiB (NSA = 3) or 128 KiB.
//    Rows are 128 B (one K-tile of one row = one full cache line per LDS-DMA row piece), 16-byte chunk c of row r sits in
//    slot c ^ ((r >> 1) & 7): applied to the per-lane SOURCE address of the lane-linear LDS-DMA image and to the read address.
//  * Global -> LDS: buffer_load_dwordx4 ... lds (1 KiB per wave instruction): per-lane voffset (row, swizzled chunk; computed
//    once), the K-tile offset in the scalar soffset - no vector address arithmetic in the loop.  8 pieces per unit and wave.
//  * K-tile t = two units (k-steps) of 64 MFMAs:
//        unit (t,0): s_barrier; MFMAs on set S0 [k-step 0]; between the 8-MFMA groups: the 8 W pieces of tile t+2 -> W stage t%2;
//                    vmcnt(8 (NSA-1)): tile t+1 has landed
//        unit (t,1): s_barrier; MFMAs on set S1|S2 [k-step 1]; between the groups: the 8 A pieces of tile t+NSA -> A stage t%NSA,
//                    and the 32 fragment reads of tile t+1 (k-step 0 -> S0, k-step 1 -> the set the previous tile used);
//                    lgkmcnt(0)
//    Hazards: a stage is overwritten only after a barrier that every wave reaches with its reads of that stage complete
//    (the lgkmcnt(0) closing unit (t-1,1) precedes the barrier opening (t,0)); a stage is read only after the barrier that follows
//    every wave's counted vmcnt for it.  Every piece has >= 2 units (NSA = 3: A pieces >= 3) between issue and first use.
#include <algorithm>

#include "ce_common.h"
#include "ce_gemm_epi.h"

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int TILE = 256 * BK * 2;  // one operand K-tile, 32 KiB
constexpr int CROW = BN * 2 + 16;   // padded epilogue staging row (528 B)
constexpr int QROW = 128 * 2 + 16;  // padded staging row of a quadrant (split-K reduce)

typedef __attribute__((address_space(3))) void lds_void;

#define W4_VM(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#define W4_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define W4_BAR() __builtin_amdgcn_s_barrier()
#define W4_PIN() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ void tile_origin_w4(int wg, int tiles_m, int tiles_n, int bn, int& m0, int& n0) {
  constexpr int GROUP = 4;  // same raster as ce_gemm256.hip (the split-K reduce of either kernel must agree with its producer)
  const int group_sz = GROUP * tiles_n;
  const int gid = wg / group_sz;
  const int first_m = gid * GROUP;
  const int gm = min(tiles_m - first_m, GROUP);
  m0 = (first_m + (wg % group_sz) % gm) * BM;
  n0 = ((wg % group_sz) / gm) * bn;
}

template <int NG>
struct FragSet {
  bf16x8 a[8], b[NG];  // one k-step (32 deep) of a 128 x 16 NG wave tile: 8 row fragments, NG column fragments (64 VGPRs at NG = 8)
};

// One MFMA of row fragment F against column fragment G.  The W fragment is the first operand, so the accumulator holds C^T:
// lane (fr, fg) of acc[F][G] owns row F*16 + fr and the four consecutive columns G*16 + fg*4 + [0,4).
// asm: D tied to C in the accumulator file ("+a").  As a builtin hipcc picks the untied form and, with all 256 AGPRs holding
// accumulators, shuffles every result back through VGPRs (hundreds of v_accvgpr_* per K-tile).  Hazards: the operands come from
// ds_reads the compiler waits for; consecutive MFMAs never share an accumulator; the first reader of the accumulators after the
// loop sits behind explicit wait states.
// REGEPI (round 6; the 256 x 256 tile, NG == 8): the A fragment is the FIRST operand instead - the accumulator
// holds C (lane (fr, fg): rows 16 F + 4 fg + [0,4), the output column of W fragment row fr) and, with the W rows permuted on the DMA's source
// side, a lane's eight G accumulators of a row are eight consecutive columns: the register-direct epilogue below.  Same products, same sums.
#define W4_MMA(F, G, S)                                                                                                      \
  if ((G) < NG) {                                                                                                            \
    if constexpr (REGEPI)                                                                                                    \
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[F][(G) < NG ? (G) : 0]) : "v"((S).a[F]), "v"((S).b[(G) < NG ? (G) : 0])); \
    else                                                                                                                     \
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[F][(G) < NG ? (G) : 0]) : "v"((S).b[(G) < NG ? (G) : 0]), "v"((S).a[F])); \
  }

// SEG2: the A operand is K-segmented on TWO nested levels (`ce_gemm256w4_seg2_launch`: the taps of a 3 x 3 x 3 convolution over
// channels-last frames - kw runs on contiguously, kh jumps a pixel row, kt a frame; ce_conv.hip) - one more scalar multiply per K-tile.
// NG: column fragments per wave - 8 = the 256 x 256 tile; 4 = a 256 x 128 tile (wave tile 128 x 64, W stages of 16 KiB) for
// products whose N is a small multiple of 128 or below it (the 96-channel convolutions of the VAE).
template <int EPI, int NSA, bool ONEBAR, bool SEG2 = false, int NG = 8>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_bf16_w4(
    const bf16* __restrict__ A, const bf16* __restrict__ W, bf16* __restrict__ C, const float* __restrict__ bias,
    const float* __restrict__ gate, const bf16* __restrict__ res, int M, int N, int K, int lda, int ldw, int ldc, int ldres,
    int gate_rows, int tiles_m, int tiles_n, int t_full, int split, float* __restrict__ ws, uint32_t a_seg_magic,
    uint32_t a_seg_extra, uint32_t w_seg_magic, uint32_t w_seg_extra, uint32_t a_seg2_magic, uint32_t a_seg2_extra) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int BN = 32 * NG;          // tile width
  constexpr bool REGEPI = NG == 8;      // register-direct epilogue (round 6; the 256 x 256 tile, plain or segmented operands - the wide VAE convs too): see W4_MMA
  constexpr int NW = NG;               // W pieces (32 rows each) per K-tile and wave
  constexpr int WTILE = BN * BK * 2;   // one W K-tile stage
  constexpr int CROW = BN * 2 + 16;    // padded epilogue staging row
  constexpr int W_RING = NSA * TILE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fg = lane >> 4;

  const bool partial = (int)blockIdx.x >= t_full;
  int wg, kt0 = 0, ktn = K / BK;
  if (!partial) {
    wg = xcd_remap(blockIdx.x, t_full);
  } else {
    const int tb = blockIdx.x - t_full;
    wg = t_full + tb / split;
    ktn = ktn / split;
    kt0 = (tb % split) * ktn;
  }
  int m0, n0;
  tile_origin_w4(wg, tiles_m, tiles_n, BN, m0, n0);
  const int kt_last = ktn - 1;

  // LDS-DMA sources: piece p of this wave = rows 8 (wave + 4 p) .. + 8 of the operand tile, lane l -> row + (l >> 3), slot l & 7
  uint32_t a_voff[8], w_voff[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int row = 8 * (wave + 4 * p) + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    a_voff[p] = (uint32_t)min(m0 + row, M - 1) * (uint32_t)(lda * 2) + chunk * 16;
    // REGEPI: LDS row 16 G + i of a wave's 128 W rows (fragment G, fragment row i) holds W row 8 i + G (whole 128-byte rows: free on the source side)
    const int wrow = REGEPI ? ((row & 128) | ((row & 15) << 3) | ((row & 127) >> 4)) : row;
    w_voff[p] = (uint32_t)min(n0 + wrow, N - 1) * (uint32_t)(ldw * 2) + chunk * 16;
  }
  const auto a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, 0xffffffffu, 0x00020000);
  const auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, 0xffffffffu, 0x00020000);
  // byte offset of K-tile t of an operand (clamped: surplus prefetches re-read the last tile; segmented operands: see ce_gemm256.hip)
  auto koff = [&](int t, uint32_t magic, uint32_t extra) __attribute__((always_inline)) -> int {
    const int ta = kt0 + min(t, kt_last);
    return ta * (BK * 2) + (int)((((uint32_t)ta * magic) >> 16) * extra);
  };
  auto koff_a = [&](int t) __attribute__((always_inline)) -> int {
    int o = koff(t, a_seg_magic, a_seg_extra);
    if (SEG2) o += (int)((((uint32_t)(kt0 + min(t, kt_last)) * a_seg2_magic) >> 16) * a_seg2_extra);
    return o;
  };
  auto dma_a = [&](int p, int stage_bytes, int soff) __attribute__((always_inline)) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_void*)(smem + stage_bytes + (wave + 4 * p) * 1024), 16, a_voff[p], soff, 0, 0);
  };
  auto dma_w = [&](int p, int stage_bytes, int soff) __attribute__((always_inline)) {
    if (p < NW)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void*)(smem + W_RING + stage_bytes + (wave + 4 * p) * 1024), 16, w_voff[p < NW ? p : 0], soff, 0, 0);
  };

  // fragment read addresses: row (wm|wn)*128 + f*16 + fr, chunk (fg + 4 ks) ^ ((row >> 1) & 7) = (fg + 4 ks) ^ (fr >> 1);  + stage + f*2048
  int a_rd[2], w_rd[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    a_rd[ks] = (wm * 128 + fr) * 128 + (((fg + 4 * ks) ^ (fr >> 1)) << 4);
    w_rd[ks] = W_RING + (wn * (BN / 2) + fr) * 128 + (((fg + 4 * ks) ^ (fr >> 1)) << 4);
  }
  // read number r (0..31) of a tile: k-step r >> 4, operand (r >> 3) & 1 (A, then W), fragment r & 7
  auto read_frag = [&](int r, int a_stage_bytes, int w_stage_bytes, FragSet<NG>& k0, FragSet<NG>& k1) __attribute__((always_inline)) {
    FragSet<NG>& s = (r >> 4) ? k1 : k0;
    const int ks = r >> 4, f = r & 7;
    if (((r >> 3) & 1) == 0)
      s.a[f] = *reinterpret_cast<const bf16x8*>(smem + a_rd[ks] + a_stage_bytes + f * 2048);
    else if (f < NG)
      s.b[f < NG ? f : 0] = *reinterpret_cast<const bf16x8*>(smem + w_rd[ks] + w_stage_bytes + f * 2048);
  };

  f32x4 acc[8][NG];
#pragma unroll
  for (int f = 0; f < 8; ++f)
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[f][g] = f32x4{0.f, 0.f, 0.f, 0.f};

  // prologue: tiles 0, 1 (and A of tile 2)
  {
    const int a0 = koff_a(0), w0 = koff(0, w_seg_magic, w_seg_extra);
    const int a1 = koff_a(1), w1 = koff(1, w_seg_magic, w_seg_extra);
#pragma unroll
    for (int p = 0; p < 8; ++p) dma_a(p, 0, a0);
#pragma unroll
    for (int p = 0; p < 8; ++p) dma_w(p, 0, w0);
#pragma unroll
    for (int p = 0; p < 8; ++p) dma_a(p, TILE, a1);
#pragma unroll
    for (int p = 0; p < 8; ++p) dma_w(p, WTILE, w1);
    if (NSA == 3 && !ONEBAR) {
      const int a2 = koff_a(2);
#pragma unroll
      for (int p = 0; p < 8; ++p) dma_a(p, 2 * TILE, a2);
    }
  }
  if (NSA == 3 && !ONEBAR) W4_VM(24); else if (NG == 8) W4_VM(16); else if (NG == 4) W4_VM(12); else W4_VM(11);  // tile 0 has landed (8 + NG pieces of tile 1 may be in flight)
  W4_BAR();
  FragSet<NG> s0, s1, s2;
#pragma unroll
  for (int r = 0; r < 32; ++r) read_frag(r, 0, 0, s0, s1);
  W4_LGKM0();
  W4_PIN();

  int a_st = 0;  // A ring stage (bytes) of the tile being multiplied; the stage of tile t + 1 is the next one, tile t + NSA lands in a_st
  // one K-tile: k-step 0 on K0, k-step 1 on K1; the next tile's fragments go to K0 (k-step 0) and KN (k-step 1)
#define W4_TILE(T, WP, K0, K1, KN)                                                                              \
  {                                                                                                             \
    const int a_next = (a_st + TILE == NSA * TILE) ? 0 : a_st + TILE;                                          \
    const int a_nn = (a_next + TILE == NSA * TILE) ? 0 : a_next + TILE;                                        \
    const int wsoff = koff((T) + 2, w_seg_magic, w_seg_extra);                                                 \
    const int asoff = koff_a((T) + (ONEBAR ? 2 : NSA));                                                        \
    const int a_dst = ONEBAR ? a_nn : a_st;                                                                     \
    if (!ONEBAR) W4_BAR();                                                                                      \
    W4_UNIT0(0, K0, WP, wsoff) W4_UNIT0(1, K0, WP, wsoff) W4_UNIT0(2, K0, WP, wsoff) W4_UNIT0(3, K0, WP, wsoff) \
    W4_UNIT0(4, K0, WP, wsoff) W4_UNIT0(5, K0, WP, wsoff) W4_UNIT0(6, K0, WP, wsoff) W4_UNIT0(7, K0, WP, wsoff) \
    if (NSA == 3 && !ONEBAR) W4_VM(16); else if (NG == 8) W4_VM(8); else if (NG == 4) W4_VM(4); else W4_VM(3);   \
    W4_BAR();                                                                                                   \
    W4_UNIT1(0, K1, K0, KN, WP) W4_UNIT1(1, K1, K0, KN, WP) W4_UNIT1(2, K1, K0, KN, WP) W4_UNIT1(3, K1, K0, KN, WP) \
    W4_UNIT1(4, K1, K0, KN, WP) W4_UNIT1(5, K1, K0, KN, WP) W4_UNIT1(6, K1, K0, KN, WP) W4_UNIT1(7, K1, K0, KN, WP) \
    W4_LGKM0();                                                                                                 \
    W4_PIN();                                                                                                   \
    a_st = a_next;                                                                                              \
  }
  // one wave per SIMD: nothing else covers an issue slot, so the fillers go BETWEEN single MFMAs (16 cycles of matrix pipe
  // each = the MFMA's own issue + about three more slots), never bunched behind a group
#define W4_UNIT0(F, K0, WP, WSOFF)                                                                                 \
  W4_MMA(F, 0, K0);                                                                                                \
  if (ONEBAR) dma_a(F, a_dst, asoff); else dma_w(F, (WP) * WTILE, WSOFF);                                          \
  W4_PIN();                                                                                                        \
  W4_MMA(F, 1, K0); W4_MMA(F, 2, K0); W4_MMA(F, 3, K0); W4_MMA(F, 4, K0); W4_MMA(F, 5, K0); W4_MMA(F, 6, K0);      \
  W4_MMA(F, 7, K0); W4_PIN();
#define W4_UNIT1(F, K1, K0, KN, WP)                                                                                \
  W4_MMA(F, 0, K1);                                                                                                \
  if (ONEBAR) dma_w(F, (WP) * WTILE, wsoff); else dma_a(F, a_dst, asoff);                                          \
  W4_PIN();                                                                                                        \
  W4_MMA(F, 1, K1); W4_PIN();                                                                                      \
  W4_MMA(F, 2, K1); read_frag(4 * (F) + 0, a_next, (1 - (WP)) * WTILE, K0, KN); W4_PIN();                           \
  W4_MMA(F, 3, K1); read_frag(4 * (F) + 1, a_next, (1 - (WP)) * WTILE, K0, KN); W4_PIN();                           \
  W4_MMA(F, 4, K1); read_frag(4 * (F) + 2, a_next, (1 - (WP)) * WTILE, K0, KN); W4_PIN();                           \
  W4_MMA(F, 5, K1); read_frag(4 * (F) + 3, a_next, (1 - (WP)) * WTILE, K0, KN); W4_PIN();                           \
  W4_MMA(F, 6, K1); W4_MMA(F, 7, K1); W4_PIN();

  const int npairs = ktn >> 1;
  for (int it = 0; it < npairs; ++it) {
    const int t = 2 * it;
    W4_TILE(t, 0, s0, s1, s2)
    W4_TILE(t + 1, 1, s0, s2, s1)
  }
#undef W4_TILE
#undef W4_UNIT0
#undef W4_UNIT1
  W4_VM(0);  // surplus prefetches must retire before the epilogue reuses the LDS
  W4_LGKM0();
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the last MFMAs' results -> v_accvgpr_read (hipcc does not see the asm MFMAs)
  static_assert(NG == 8 || ((NG == 4 || NG == 3) && NSA == 2 && !ONEBAR), "the 256 x 128 / 256 x 96 tiles exist for the two-stage loop only");
  if constexpr (REGEPI) {
    // ---- epilogue, REGISTER-DIRECT (round 6; as csrc/ce_gemm384.hip and csrc/ce_gemm_fp8w4.hip): lane (fr, fg) of acc[f][g] owns rows f*16 + fg*4 +
    // [0,4) of the wave tile and output column 8 fr + g - eight consecutive columns per row over its eight g accumulators: one 16-byte store per
    // lane, 256 contiguous bytes of a row per quad-row of lanes, four rows per wave instruction; no LDS round trip, no barrier.  Bit-identical
    // to the staged form (same arithmetic, same roundings).
    const int col0 = n0 + wn * 128 + fr * 8;
    const bool col_ok = col0 < N;
    const int colc = min(col0, N - 8);
    const int row_base = m0 + wm * 128 + fg * 4;  // + f*16 + jj
    if (partial) {
      // split-K tail piece: fp32 slab in the [wave][f][g][lane'] order gemm256w4_reduce reads (the C^T accumulator order): this lane's
      // (f, jj, g = 4 h .. 4 h + 3) -> lane' = fg*4 + jj + 16 (2 (fr & 1) + h) of [f][fr >> 1]
      float* slab = ws + (size_t)(blockIdx.x - t_full) * (BM * BN);
#pragma unroll
      for (int f = 0; f < 8; ++f) {
        f32x4 av[8];
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          asm volatile("" : "+a"(acc[f][g]));  // (pins the read-out of fragment row f here)
          av[g] = acc[f][g];
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const f32x4 o = {av[4 * h + 0][jj], av[4 * h + 1][jj], av[4 * h + 2][jj], av[4 * h + 3][jj]};
            const int lane_o = fg * 4 + jj + 16 * (2 * (fr & 1) + h);
            *reinterpret_cast<f32x4*>(slab + (((wave * 64 + f * 8 + (fr >> 1)) * 64) + lane_o) * 4) = o;
          }
      }
      return;
    }
    f32x4 bvv[2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
      bvv[h] = (EPI != EPI_BIAS_ROW && bias != nullptr) ? *reinterpret_cast<const f32x4*>(bias + colc + 4 * h) : f32x4{0.f, 0.f, 0.f, 0.f};
    // (the launcher sends C beyond 32-bit byte offsets, gate rows shorter than a tile and row biases with M % 4 != 0 to the 8-wave kernel)
    const auto c_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)C, 0, (uint32_t)(M - 1) * (uint32_t)(ldc * 2) + (uint32_t)N * 2u, 0x00020000);
    f32x4 gA[2], gB[2];
    int g_switch = 0x7fffffff;
    u32x4 rv[8][4];  // gated residual: all 32 residual pieces of this lane requested in front of the first row (the fragment registers are free)
    if (EPI == EPI_GATE_RES) {
      gA[0] = gA[1] = gB[0] = gB[1] = f32x4{1.f, 1.f, 1.f, 1.f};
      if (gate != nullptr) {
        const int s0 = gate_rows > 0 ? m0 / gate_rows : 0;
        const int s1 = gate_rows > 0 ? min(M - 1, m0 + BM - 1) / gate_rows : 0;
        const float* ga = gate + (size_t)s0 * N + colc;
        const float* gb = gate + (size_t)s1 * N + colc;
        gA[0] = *reinterpret_cast<const f32x4*>(ga);
        gA[1] = *reinterpret_cast<const f32x4*>(ga + 4);
        gB[0] = *reinterpret_cast<const f32x4*>(gb);
        gB[1] = *reinterpret_cast<const f32x4*>(gb + 4);
        if (s1 != s0) g_switch = s1 * gate_rows;
      }
#pragma unroll
      for (int f = 0; f < 8; ++f)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
          rv[f][jj] = *reinterpret_cast<const u32x4*>(res + (size_t)min(row_base + f * 16 + jj, M - 1) * ldres + colc);
    }
#pragma unroll
    for (int f = 0; f < 8; ++f) {
      f32x4 av[8];
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        asm volatile("" : "+a"(acc[f][g]));  // (pins the read-out of fragment row f here)
        av[g] = acc[f][g];
      }
      f32x4 brow4 = {0.f, 0.f, 0.f, 0.f};
      if (EPI == EPI_BIAS_ROW) brow4 = *reinterpret_cast<const f32x4*>(bias + min(row_base + f * 16, M - 4));
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int m = row_base + f * 16 + jj;
        u32x4 y;  // bf16(acc + bias), 8 columns
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int h = q >> 1, e = 2 * (q & 1);
          const float b0 = EPI == EPI_BIAS_ROW ? brow4[jj] : bvv[h][e], b1 = EPI == EPI_BIAS_ROW ? brow4[jj] : bvv[h][e + 1];
          y[q] = pack_bf16(av[2 * q][jj] + b0, av[2 * q + 1][jj] + b1);
        }
        u32x4 o = y;
        if (EPI == EPI_BIAS_GELU) {
#pragma unroll
          for (int q = 0; q < 4; ++q) o[q] = pack_bf16(gelu_tanh(bf16lo(y[q])), gelu_tanh(bf16hi(y[q])));
        } else if (EPI == EPI_BIAS_GELU_ERF) {
#pragma unroll
          for (int q = 0; q < 4; ++q) o[q] = pack_bf16(gelu_erf(bf16lo(y[q])), gelu_erf(bf16hi(y[q])));
        } else if (EPI == EPI_GATE_RES) {
          const u32x4 r = rv[f][jj];
          const bool second = m >= g_switch;
          const f32x4 g0 = second ? gB[0] : gA[0], g1 = second ? gB[1] : gA[1];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float ga = q < 2 ? g0[2 * q] : g1[2 * q - 4], gb = q < 2 ? g0[2 * q + 1] : g1[2 * q - 3];
            // x.float() + y * gate with both fp32 roundings of the reference (transformer_chronoedit.py:281,293): no fma contraction
            o[q] = pack_bf16(mul_then_add(bf16lo(y[q]), ga, bf16lo(r[q])), mul_then_add(bf16hi(y[q]), gb, bf16hi(r[q])));
          }
        }
        const uint32_t coff = (m < M && col_ok) ? (uint32_t)m * (uint32_t)(ldc * 2) + (uint32_t)col0 * 2u : 0xffffffffu;
        __builtin_amdgcn_raw_buffer_store_b128(o, c_rsrc, coff, 0, 0);
      }
    }
    return;
  }
  W4_BAR();

  if (partial) {  // split-K tail piece: fp32 slab [wave][f][g][lane] for gemm256w4_reduce (the launcher splits 256 x 256 tiles only)
    float* slab = ws + (size_t)(blockIdx.x - t_full) * (BM * BN);
#pragma unroll
    for (int f = 0; f < 8; ++f)
#pragma unroll
      for (int g = 0; g < NG; ++g)
        *reinterpret_cast<f32x4*>(slab + (((wave * 64 + f * 8 + g) * 64) + lane) * 4) = acc[f][g];
    return;
  }
  constexpr int CPR = BN / 8;        // 16-byte chunks per staged row (32 | 16 | 12)
  // chunks per thread and pass: 8 | 4 when a staged row's chunk count divides 256 (thread t: chunk t % CPR of rows t / CPR + (256 / CPR) tt);
  // the 96-wide tile (CPR = 12): threads 0..251 take chunk t % 12 of rows t / 12 + 21 tt, tt = 0..3 - rows past 63 and threads 252..255 idle
  constexpr bool CPOW2 = (CPR & (CPR - 1)) == 0;
  constexpr int RPT = 256 / CPR;                        // staged rows one sweep of the workgroup covers (8 | 16 | 21)
  constexpr int CH = CPOW2 ? 64 * CPR / 256 : (64 + RPT - 1) / RPT;
  const int my_cc = CPOW2 ? (tid & (CPR - 1)) : tid % CPR;
  // staged row of this thread's chunk tt (clamped to a valid one) and whether the chunk exists
  auto chunk_row = [&](int tt, bool& ok) __attribute__((always_inline)) -> int {
    if (CPOW2) {
      ok = true;
      return (tid + 256 * tt) / CPR;
    }
    const int rl = tid / CPR + RPT * tt;
    ok = tid < RPT * CPR && rl < 64;
    return min(rl, 63);
  };

  // ---- epilogue: four passes of 64 staged rows (pass p: accumulator rows f = 2p, 2p+1 of every wave = tile rows
  // wm*128 + p*32 + [0,32)), so that bias / GELU / gated-residual math and the global stores run on 16-B row-contiguous chunks
  f32x4 bcol[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int n = n0 + wn * (BN / 2) + g * 16 + fg * 4;
    bcol[g] = (EPI != EPI_BIAS_ROW && bias != nullptr) ? *reinterpret_cast<const f32x4*>(bias + min(n, N - 4)) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // Gated residual: ALL of this thread's residual chunks (4 passes x 8 chunks x 16 B = 128 VGPRs - the fragment registers are
  // free now) and its gate values are requested HERE, before the first staging pass, so the four passes below run without a
  // memory round trip each (as per-pass loads the epilogue of a K = 5120 tile cost ~12 us of its ~120 us: four serial HBM
  // latencies).  The thread that reads a chunk is the thread that stores it (res may alias C).  The gate of a thread is one
  // column chunk of at most two samples' rows when gate_rows >= the tile height (the engine: tokens per sample); other callers
  // (gate_rows < 256) take the per-pass path of ce_gemm_epi.h.
  constexpr bool prefetch = EPI == EPI_GATE_RES;  // (the launcher sends 0 < gate_rows < 256 to the 8-wave kernel)
  u32x4 rv[4][CH];
  f32x4 gA0, gA1, gB0, gB1;
  int g_switch = 0x7fffffff;  // first global row that takes the second sample's gate
  const int my_n = n0 + my_cc * 8, my_nc = min(my_n, N - 8);
  const auto c_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)C, 0, (uint32_t)(M - 1) * (uint32_t)(ldc * 2) + (uint32_t)N * 2u, 0x00020000);
  if (EPI == EPI_GATE_RES && prefetch) {
    gA0 = gA1 = gB0 = gB1 = f32x4{1.f, 1.f, 1.f, 1.f};
    if (gate != nullptr) {  // first: in-order returns, and the first chunk needs them
      const int s0 = gate_rows > 0 ? m0 / gate_rows : 0;
      const int s1 = gate_rows > 0 ? min(M - 1, m0 + BM - 1) / gate_rows : 0;
      const float* ga = gate + (size_t)s0 * N + my_nc;
      const float* gb = gate + (size_t)s1 * N + my_nc;
      gA0 = *reinterpret_cast<const f32x4*>(ga);
      gA1 = *reinterpret_cast<const f32x4*>(ga + 4);
      gB0 = *reinterpret_cast<const f32x4*>(gb);
      gB1 = *reinterpret_cast<const f32x4*>(gb + 4);
      if (s1 != s0) g_switch = s1 * gate_rows;
    }
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int tt = 0; tt < CH; ++tt) {
        bool ok;
        const int rl = chunk_row(tt, ok);
        const int m = min(m0 + (rl >> 5) * 128 + p * 32 + (rl & 31), M - 1);
        rv[p][tt] = *reinterpret_cast<const u32x4*>(res + (size_t)m * ldres + my_nc);
      }
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    if (p > 0) __syncthreads();
#pragma unroll
    for (int ff = 0; ff < 2; ++ff) {
      const int f = 2 * p + ff;
      float brow = 0.f;
      if (EPI == EPI_BIAS_ROW) brow = bias[min(m0 + wm * 128 + f * 16 + fr, M - 1)];
      const int rl = wm * 32 + ff * 16 + fr;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int cl = wn * (BN / 2) + g * 16 + fg * 4;
        f32x4 bv = bcol[g];
        if (EPI == EPI_BIAS_ROW) bv[0] = bv[1] = bv[2] = bv[3] = brow;
        const f32x4 v = acc[f][g];
        const u32x2 pk = {pack_bf16(v[0] + bv[0], v[1] + bv[1]), pack_bf16(v[2] + bv[2], v[3] + bv[3])};
        *reinterpret_cast<u32x2*>(smem + rl * CROW + cl * 2) = pk;
      }
    }
    __syncthreads();
    if (EPI == EPI_GATE_RES && prefetch) {
#pragma unroll
      for (int tt = 0; tt < CH; ++tt) {
        bool ok;
        const int rl = chunk_row(tt, ok);
        const int m = ok ? m0 + (rl >> 5) * 128 + p * 32 + (rl & 31) : M;  // (a chunk that does not exist: its store falls outside the buffer's range)
        const u32x4 y = *reinterpret_cast<const u32x4*>(smem + rl * CROW + my_cc * 16);
        const bool second = m >= g_switch;
        const f32x4 g0 = second ? gB0 : gA0, g1 = second ? gB1 : gA1;
        const u32x4 r = rv[p][tt];
        u32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float ga = q < 2 ? g0[2 * q] : g1[2 * q - 4], gb = q < 2 ? g0[2 * q + 1] : g1[2 * q - 3];
          // x.float() + y * gate with both fp32 roundings of the reference (transformer_chronoedit.py:281,293): no fma contraction
          o[q] = pack_bf16(mul_then_add(bf16lo(y[q]), ga, bf16lo(r[q])), mul_then_add(bf16hi(y[q]), gb, bf16hi(r[q])));
        }
        // predicated by the buffer's range check, not by a branch: hipcc sinks a chunk's residual load into a branch that holds its
        // only use (one exposed memory round trip per pass again)
        const uint32_t coff = (m < M && my_n < N) ? (uint32_t)m * (uint32_t)(ldc * 2) + (uint32_t)my_n * 2u : 0xffffffffu;
        __builtin_amdgcn_raw_buffer_store_b128(o, c_rsrc, coff, 0, 0);
      }
    } else if (EPI != EPI_GATE_RES) {
      epi_chunks<EPI, CH>(smem, CROW,
                         [&](int tt, int& rl, int& cc, int& mr) {
                           bool ok;
                           rl = chunk_row(tt, ok);
                           cc = my_cc;
                           mr = ok ? (rl >> 5) * 128 + p * 32 + (rl & 31) : (1 << 28);  // (no such chunk: row past M, the store is skipped)
                         },
                         m0, n0, C, gate, res, M, N, ldc, ldres, gate_rows);
    }
  }
}

// Sums the `split` fp32 slabs of one quadrant (= one wave's 128 x 128 accumulators) of a tail tile and applies the epilogue.
// grid = 4 x the number of tail tiles, 256 threads: thread (w, lane) takes accumulator rows f = 2w, 2w+1 of the quadrant.
template <int EPI>
__global__ __launch_bounds__(256) void gemm256w4_reduce(bf16* __restrict__ C, const float* __restrict__ bias, const float* __restrict__ gate,
                                                        const bf16* __restrict__ res, int M, int N, int ldc, int ldres, int gate_rows,
                                                        int tiles_m, int tiles_n, int t_full, int split, const float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int tile = blockIdx.x >> 2, q = blockIdx.x & 3;  // q = producer wave = (wm, wn)
  int m0, n0;
  tile_origin_w4(t_full + tile, tiles_m, tiles_n, BN, m0, n0);
  m0 += (q >> 1) * 128;
  n0 += (q & 1) * 128;
  const float* slab = ws + (size_t)tile * split * (BM * BN);
#pragma unroll
  for (int ff = 0; ff < 2; ++ff) {
    const int f = 2 * w + ff;
    const int rl = f * 16 + fr;
    float brow = 0.f;
    if (EPI == EPI_BIAS_ROW) brow = bias[min(m0 + rl, M - 1)];
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const int cl = g * 16 + fg * 4;
      f32x4 bv = (EPI != EPI_BIAS_ROW && bias != nullptr) ? *reinterpret_cast<const f32x4*>(bias + min(n0 + cl, N - 4)) : f32x4{0.f, 0.f, 0.f, 0.f};
      if (EPI == EPI_BIAS_ROW) bv[0] = bv[1] = bv[2] = bv[3] = brow;
      const int e = (((q * 64 + f * 8 + g) * 64) + lane) * 4;
      f32x4 v = *reinterpret_cast<const f32x4*>(slab + e);
      for (int sidx = 1; sidx < split; ++sidx) v += *reinterpret_cast<const f32x4*>(slab + (size_t)sidx * (BM * BN) + e);
      const u32x2 pk = {pack_bf16(v[0] + bv[0], v[1] + bv[1]), pack_bf16(v[2] + bv[2], v[3] + bv[3])};
      *reinterpret_cast<u32x2*>(smem + rl * QROW + cl * 2) = pk;
    }
  }
  __syncthreads();
  epi_chunks<EPI, 8>(smem, QROW, [&](int tt, int& rl, int& cc, int& mr) { const int c = tid + 256 * tt; rl = mr = c >> 4; cc = c & 15; }, m0, n0,
                     C, gate, res, M, N, ldc, ldres, gate_rows);
}

}  // namespace

extern "C" void ce_gemm256_workspace(hipStream_t stream, float** ws, size_t* bytes, int* cus);

// nsa: 3 = A ring of three K-tile stages (160 KiB of LDS), 2 = two (128 KiB), 1 = three stages and ONE barrier per K-tile
extern "C" int ce_gemm256_launch(const void* A, const void* W, void* C, const float* bias, int epilogue, const float* gate,
                                 const void* res, int M, int N, int K, int lda, int ldw, int ldc, int ldres, int gate_rows,
                                 int a_seg_k, long long a_seg_stride, int w_seg_k, long long w_seg_stride, hipStream_t stream);

static int w4_launch(const void* A, const void* W, void* C, const float* bias, int epilogue, const float* gate, const void* res, int M, int N,
                     int K, int lda, int ldw, int ldc, int ldres, int gate_rows, int a_seg_k, long long a_seg_stride, int w_seg_k,
                     long long w_seg_stride, int a_seg2_k, long long a_seg2_stride, int nsa, int ng, hipStream_t stream) {
  const bool seg2 = a_seg2_k > 0;
  if ((ng != 8 && ng != 4 && ng != 3) || (ng != 8 && !seg2)) return CE_ERR_ARG;
  if (seg2 && (a_seg_k <= 0 || a_seg2_k % a_seg_k || (epilogue != EPI_BIAS && epilogue != EPI_GATE_RES) || nsa != 2 ||
               (epilogue == EPI_GATE_RES && gate != nullptr)))
    return CE_ERR_ARG;
  // the prefetched gated-residual epilogue holds ONE or TWO samples' gate rows per tile: gate rows shorter than a tile -> 8-wave kernel
  // ... and stores through a 32-bit-offset buffer descriptor
  if (epilogue == EPI_GATE_RES && ((gate != nullptr && gate_rows > 0 && gate_rows < BM) || (long long)M * ldc * 2 >= (1ll << 32)))
    return ce_gemm256_launch(A, W, C, bias, epilogue, gate, res, M, N, K, lda, ldw, ldc, ldres, gate_rows, a_seg_k, a_seg_stride, w_seg_k,
                             w_seg_stride, stream);
  // (the register-direct epilogue of the plain 256 x 256 tile stores through 32-bit buffer offsets and reads four row biases at once)
  if (ng == 8 && ((long long)M * ldc * 2 >= (1ll << 32) || (epilogue == EPI_BIAS_ROW && (M & 3))))
    return ce_gemm256_launch(A, W, C, bias, epilogue, gate, res, M, N, K, lda, ldw, ldc, ldres, gate_rows, a_seg_k, a_seg_stride, w_seg_k,
                             w_seg_stride, stream);
  const int bn = 32 * ng;
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + bn - 1) / bn;
  const int nwg = tiles_m * tiles_n, kt = K / BK;
  uint32_t a_seg_magic = 0, a_seg_extra = 0, w_seg_magic = 0, w_seg_extra = 0;
  auto seg = [&](int seg_k, long long seg_stride, uint32_t& magic, uint32_t& extra_out) -> int {
    if (seg_k <= 0 || seg_k >= K) return CE_OK;
    if (seg_k % BK) return CE_ERR_SHAPE;
    const int tps = seg_k / BK;
    magic = 65536u / (uint32_t)tps + 1u;
    for (int t = 0; t < kt; ++t)
      if ((int)(((uint32_t)t * magic) >> 16) != t / tps) return CE_ERR_SHAPE;
    const long long extra = (seg_stride - seg_k) * 2;
    if (extra < 0 || extra * ((K + seg_k - 1) / seg_k) + (long long)K * 2 >= (1ll << 31)) return CE_ERR_SHAPE;  // the K-tile offset is a signed scalar
    extra_out = (uint32_t)extra;
    return CE_OK;
  };
  if (int rc = seg(a_seg_k, a_seg_stride, a_seg_magic, a_seg_extra)) return rc;
  if (int rc = seg(w_seg_k, w_seg_stride, w_seg_magic, w_seg_extra)) return rc;
  uint32_t a_seg2_magic = 0, a_seg2_extra = 0;
  if (seg2 && a_seg2_k < K) {  // second level: every a_seg2_k columns the source jumps to a_seg2_stride (both in elements), nested in the first
    if (a_seg2_k % BK) return CE_ERR_SHAPE;
    const int tps = a_seg2_k / BK;
    a_seg2_magic = 65536u / (uint32_t)tps + 1u;
    for (int t = 0; t < kt; ++t)
      if ((int)(((uint32_t)t * a_seg2_magic) >> 16) != t / tps) return CE_ERR_SHAPE;
    const long long extra = (a_seg2_stride - (long long)(a_seg2_k / a_seg_k) * a_seg_stride) * 2;
    const long long reach = ((long long)(K - 1) / a_seg2_k) * a_seg2_stride * 2 + (long long)(a_seg2_k / a_seg_k) * a_seg_stride * 2 + (long long)a_seg_k * 2;
    if (extra < 0 || reach >= (1ll << 31)) return CE_ERR_SHAPE;
    a_seg2_extra = (uint32_t)extra;
  }
  float* g_ws = nullptr;
  size_t g_ws_bytes = 0;
  int g_cus = 256;
  ce_gemm256_workspace(stream, &g_ws, &g_ws_bytes, &g_cus);
  int tail = nwg % g_cus, split = 1;
  if (tail > 0 && g_ws != nullptr && !seg2) {
    for (int s = std::min(g_cus / tail, 8); s >= 2; --s)
      if (kt % (2 * s) == 0 && (size_t)tail * s * BM * BN * sizeof(float) <= g_ws_bytes) {
        split = s;
        break;
      }
  }
  if (split == 1) tail = 0;
  const int t_full2 = nwg - tail;
  dim3 grid(t_full2 + tail * split), block(256);
  const int lds3 = 5 * TILE, lds2 = 4 * TILE;
  if (seg2) {
    const int lds2n = 2 * TILE + 2 * (128 * BK * 2);  // 256 x 128 tile: W stages of 16 KiB
    const int lds2m = 2 * TILE + 2 * (96 * BK * 2);   // 256 x 96 tile: W stages of 12 KiB
    static bool seg2_done_[CE_MAX_DEVICES] = {};
    bool& seg2_done = seg2_done_[ce_device_slot()];
    if (!seg2_done) {
      if (hipFuncSetAttribute((const void*)gemm_bf16_w4<EPI_BIAS, 2, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2) != hipSuccess ||
          hipFuncSetAttribute((const void*)gemm_bf16_w4<EPI_GATE_RES, 2, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2) != hipSuccess ||
          hipFuncSetAttribute((const void*)gemm_bf16_w4<EPI_BIAS, 2, false, true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2n) != hipSuccess ||
          hipFuncSetAttribute((const void*)gemm_bf16_w4<EPI_GATE_RES, 2, false, true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2n) != hipSuccess ||
          hipFuncSetAttribute((const void*)gemm_bf16_w4<EPI_BIAS, 2, false, true, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2m) != hipSuccess ||
          hipFuncSetAttribute((const void*)gemm_bf16_w4<EPI_GATE_RES, 2, false, true, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2m) != hipSuccess)
        return CE_ERR_ARG;
      seg2_done = true;
    }
#define CE_LAUNCH_SEG2(E, NGV, LDS)                                                                                                     \
  hipLaunchKernelGGL((gemm_bf16_w4<E, 2, false, true, NGV>), grid, block, LDS, stream, (const bf16*)A, (const bf16*)W, (bf16*)C, bias, gate, \
                     (const bf16*)res, M, N, K, lda, ldw, ldc, ldres, gate_rows, tiles_m, tiles_n, t_full2, split, g_ws, a_seg_magic,   \
                     a_seg_extra, w_seg_magic, w_seg_extra, a_seg2_magic, a_seg2_extra)
    if (epilogue == EPI_BIAS) {
      if (ng == 8) CE_LAUNCH_SEG2(EPI_BIAS, 8, lds2); else if (ng == 4) CE_LAUNCH_SEG2(EPI_BIAS, 4, lds2n); else CE_LAUNCH_SEG2(EPI_BIAS, 3, lds2m);
    } else {
      if (ng == 8) CE_LAUNCH_SEG2(EPI_GATE_RES, 8, lds2); else if (ng == 4) CE_LAUNCH_SEG2(EPI_GATE_RES, 4, lds2n); else CE_LAUNCH_SEG2(EPI_GATE_RES, 3, lds2m);
    }
#undef CE_LAUNCH_SEG2
    return (int)hipGetLastError();
  }
  static bool attr_done_[CE_MAX_DEVICES][8] = {};
  bool* attr_done = attr_done_[ce_device_slot()];
#define CE_LAUNCH(E)                                                                                                       \
  do {                                                                                                                     \
    if (!attr_done[E]) {                                                                                                   \
      if (hipFuncSetAttribute((const void*)gemm_bf16_w4<E, 3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds3) != hipSuccess) return CE_ERR_ARG; \
      if (hipFuncSetAttribute((const void*)gemm_bf16_w4<E, 3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds3) != hipSuccess) return CE_ERR_ARG; \
      if (hipFuncSetAttribute((const void*)gemm_bf16_w4<E, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2) != hipSuccess) return CE_ERR_ARG; \
      (void)hipFuncSetAttribute((const void*)gemm256w4_reduce<E>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * QROW); \
      attr_done[E] = true;                                                                                                 \
    }                                                                                                                      \
    if (nsa == 1)                                                                                                          \
      hipLaunchKernelGGL((gemm_bf16_w4<E, 3, true>), grid, block, lds3, stream, (const bf16*)A, (const bf16*)W, (bf16*)C, bias, gate, \
                         (const bf16*)res, M, N, K, lda, ldw, ldc, ldres, gate_rows, tiles_m, tiles_n, t_full2, split, g_ws, \
                         a_seg_magic, a_seg_extra, w_seg_magic, w_seg_extra, 0u, 0u);                                      \
    else if (nsa == 3)                                                                                                     \
      hipLaunchKernelGGL((gemm_bf16_w4<E, 3, false>), grid, block, lds3, stream, (const bf16*)A, (const bf16*)W, (bf16*)C, bias, gate, \
                         (const bf16*)res, M, N, K, lda, ldw, ldc, ldres, gate_rows, tiles_m, tiles_n, t_full2, split, g_ws, \
                         a_seg_magic, a_seg_extra, w_seg_magic, w_seg_extra, 0u, 0u);                                      \
    else                                                                                                                   \
      hipLaunchKernelGGL((gemm_bf16_w4<E, 2, false>), grid, block, lds2, stream, (const bf16*)A, (const bf16*)W, (bf16*)C, bias, gate, \
                         (const bf16*)res, M, N, K, lda, ldw, ldc, ldres, gate_rows, tiles_m, tiles_n, t_full2, split, g_ws, \
                         a_seg_magic, a_seg_extra, w_seg_magic, w_seg_extra, 0u, 0u);                                      \
    if (tail)                                                                                                              \
      hipLaunchKernelGGL((gemm256w4_reduce<E>), dim3(4 * tail), block, 128 * QROW, stream, (bf16*)C, bias, gate,           \
                         (const bf16*)res, M, N, ldc, ldres, gate_rows, tiles_m, tiles_n, t_full2, split, g_ws);           \
  } while (0)
  switch (epilogue) {
    case EPI_BIAS: CE_LAUNCH(EPI_BIAS); break;
    case EPI_BIAS_GELU: CE_LAUNCH(EPI_BIAS_GELU); break;
    case EPI_GATE_RES: CE_LAUNCH(EPI_GATE_RES); break;
    case EPI_BIAS_GELU_ERF: CE_LAUNCH(EPI_BIAS_GELU_ERF); break;
    case EPI_BIAS_ROW: CE_LAUNCH(EPI_BIAS_ROW); break;
    default: return CE_ERR_ARG;
  }
#undef CE_LAUNCH
  return (int)hipGetLastError();
}

extern "C" int ce_gemm256w4_launch(const void* A, const void* W, void* C, const float* bias, int epilogue, const float* gate,
                                   const void* res, int M, int N, int K, int lda, int ldw, int ldc, int ldres, int gate_rows,
                                   int a_seg_k, long long a_seg_stride, int w_seg_k, long long w_seg_stride, int nsa, hipStream_t stream) {
  return w4_launch(A, W, C, bias, epilogue, gate, res, M, N, K, lda, ldw, ldc, ldres, gate_rows, a_seg_k, a_seg_stride, w_seg_k, w_seg_stride,
                   0, 0, nsa, 8, stream);
}

// A with TWO nested segment levels: column k of row m lives at A + (k / a_seg2_k) a_seg2_stride + ((k % a_seg2_k) / a_seg_k) a_seg_stride +
// m lda + k % a_seg_k (elements).  EPI_BIAS or the plain residual add (EPI_GATE_RES without a gate); no split-K.
// n_tile: 256, 128 (the 256 x 128 macro tile: wave tiles 128 x 64) or 96 (256 x 96: wave tiles 128 x 48 - the 96-channel convolutions).
extern "C" int ce_gemm256w4_seg2_launch(const void* A, const void* W, void* C, const float* bias, int epilogue, const void* res, int M, int N,
                                        int K, int lda, int ldw, int ldc, int ldres, int a_seg_k, long long a_seg_stride, int a_seg2_k,
                                        long long a_seg2_stride, int n_tile, hipStream_t stream) {
  return w4_launch(A, W, C, bias, epilogue, nullptr, res, M, N, K, lda, ldw, ldc, ldres, 0, a_seg_k, a_seg_stride, 0, 0, a_seg2_k, a_seg2_stride,
                   2, n_tile / 32, stream);
}

// The split-K reduce of this file's slab layout for another producer (ce_gemm_fp8w4.hip: the same wave tiles and raster, slabs already
// scaled): sums the `split` slabs of each of the `tail` tiles, adds the bias, applies the epilogue.
extern "C" int ce_gemm256w4_reduce_launch(int epilogue, void* C, const float* bias, const float* gate, const void* res, int M, int N, int ldc,
                                          int ldres, int gate_rows, int tiles_m, int tiles_n, int t_full, int split, const float* ws, int tail,
                                          hipStream_t stream) {
  static bool done_[CE_MAX_DEVICES][8] = {};
  bool* done = done_[ce_device_slot()];
#define CE_RED(E)                                                                                                            \
  do {                                                                                                                       \
    if (!done[E]) {                                                                                                          \
      (void)hipFuncSetAttribute((const void*)gemm256w4_reduce<E>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * QROW);    \
      done[E] = true;                                                                                                        \
    }                                                                                                                        \
    hipLaunchKernelGGL((gemm256w4_reduce<E>), dim3(4 * tail), dim3(256), 128 * QROW, stream, (bf16*)C, bias, gate, (const bf16*)res, M, N, \
                       ldc, ldres, gate_rows, tiles_m, tiles_n, t_full, split, ws);                                          \
  } while (0)
  switch (epilogue) {
    case EPI_BIAS: CE_RED(EPI_BIAS); break;
    case EPI_BIAS_GELU: CE_RED(EPI_BIAS_GELU); break;
    case EPI_GATE_RES: CE_RED(EPI_GATE_RES); break;
    default: return CE_ERR_ARG;
  }
#undef CE_RED
  return (int)hipGetLastError();
}
