// 256x256x128 fp8 (OCP e4m3) GEMM with per-row / per-column scales, ONE WAVE PER SIMD: 4 waves (2 x 2), wave tile 128 x 128 -
// the register / LDS economy of ce_gemm256w4.hip applied to the contract of ce_gemm_fp8.hip (same arithmetic, same epilogues):
//     C[m][n] = epilogue( sa[m] * sw[n] * sum_k Aq[m][k] * Wq[n][k] + bias[n] )
// A 128-byte LDS row is 128 fp8 = ONE k-step of v_mfma_f32_16x16x128_f8f6f4 (the unscaled form: block scales 2^0), so a K-tile is
// 64 MFMAs per wave on 16 fragments of 32 bytes per lane (8 registers each) - 32 ds_read_b128 per K-tile and wave where the 8-wave
// kernel reads 48, and one workgroup barrier per K-tile instead of four.
//
//  * registers: 64 accumulators = 256 AGPRs (asm MFMAs tied in place, as in ce_gemm256w4.hip); the W fragments of a K-tile
//    double-buffered (2 x 8 x 8 = 128 VGPRs); the A fragments through a RING of four (32 VGPRs) read three MFMA groups ahead.
//  * LDS: two stages of [A 32 KiB | W 32 KiB]; rows of 128 B with the source-side chunk swizzle (chunk c of row r in slot
//    c ^ ((r >> 1) & 7)), filled by buffer_load_dwordx4 ... lds with per-lane row offsets computed once and the K-tile in soffset.
//  * K-tile t (stage t & 1) = 8 groups; group G = A fragment G against the eight W fragments:
//        groups 0..4: the ring reads A fragments 3..7 of tile t
//        vmcnt(0) + lgkmcnt(0) + s_barrier: tile t+1 has landed everywhere, nobody reads stage t & 1 any more (its W fragments were
//                     taken during tile t-1, its last A fragment in group 4)
//        groups 5..7: the 16 LDS-DMA pieces of tile t+2 -> stage t & 1; the W fragments of tile t+1 -> the other W set; the ring
//                     wraps into tile t+1 (A fragments 0..2)
//    so a piece has five groups (>= 1280 matrix-pipe cycles) plus the wait in front of the barrier to land.
//
// MX form (round 4; template flag MX, entry ce_gemm_mxfp8): both operands carry OCP-MX block scales - one E8M0 byte per 32 consecutive K
// elements of a row - and the SCALED form of the same instruction applies them inside the matrix pipe:
//     C[m][n] = epilogue( sum_blocks 2^(ea[m][blk] + ew[n][blk]) * sum_{k in blk} Aq[m][k] * Wq[n][k] + bias[n] )
// Operand geometry of v_mfma_scale_f32_16x16x128_f8f6f4 (tools/probes/mx16_probe.hip, profiles/r04_mx16_probe.txt): lane (r, g) feeds
// row r with 32 bytes - bytes 0-15 are k = 16 g .. 16 g + 15 and bytes 16-31 are k = 64 + 16 g .. of the 128-deep step (so MX block
// beta = k / 32 is bytes 0-15 of lanes g = 2 beta, 2 beta + 1 for beta < 2 and bytes 16-31 of lanes g = 2 (beta - 2), + 1 otherwise) -
// and the scale byte of block beta is taken from lane (r, beta), byte op_sel + 2 op_sel_hi of its scale register.  So a lane fetches
// the 16-byte chunks g and 4 + g of its row's 128 bytes (the unscaled form reads 2 g and 2 g + 1: any packing common to both operands
// gives the same product) and passes the scale of block g.  Scales live in memory in the order this loop reads them,
// [row / 128][K / 128][g = 4][row % 16][(row / 16) % 8] bytes: the eight row fragments of a wave tile are 8 consecutive bytes, so ONE
// 8-byte load per lane, operand and K-tile (512 contiguous bytes per wave) brings all its scales, and op_sel picks fragment F & 3 out of
// register F >> 2 - 0.8 % of the operand bytes, fetched one K-tile ahead into 8 registers.  (A first form - [row][4][K / 128], one dword
// per fragment covering four K-tiles - needed 32 registers and spilled.)
#include <algorithm>

#include "ce_common.h"
#include "ce_gemm_epi.h"

// F8_A3 = 1: an A ring of THREE K-tile stages beside the W ring of two (160 KiB of LDS, as ce_gemm256w4.hip).  The A pieces of tile T+2 then go out in
// groups 1..4 of tile T (nine groups of flight) and the W pieces of tile T+2 in groups 5, 6, 7 of T and group 0 of T+1 (five): TWO pieces per group
// everywhere, none with less than five groups to land (the two-stage loop: three pieces per group and three groups).
#ifndef F8_A3
#define F8_A3 0
#endif
// F8_EPI_LDS = 1: the round-3..5 epilogue (accumulators hold C^T, rows staged through LDS for row-contiguous stores; W scales in the A order) -
// kept as the A/B partner of tools/gemm_mxfp8_ab.py.  0 (round 6, default): the REGISTER-DIRECT epilogue, see "epilogue" below.
#ifndef F8_EPI_LDS
#define F8_EPI_LDS 0
#endif

namespace {

constexpr int BM = 256, BN = 256, BKB = 128;  // K-tile in bytes (= fp8 elements)
constexpr int TILE = 256 * BKB;               // one operand K-tile, 32 KiB
constexpr int STAGE = 2 * TILE;               // [A | W]
constexpr int CROW = BN * 2 + 16;             // padded epilogue staging row (528 B)
constexpr int W_RING = 3 * TILE;              // (F8_A3) A stages at 0, TILE, 2 TILE; W stages at W_RING, W_RING + TILE
constexpr int LDS_BYTES = F8_A3 ? 5 * TILE : 2 * STAGE;

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((ext_vector_type(8))) int i32x8;

// LDS-DMA issue schedule of the main loop: how many of a K-tile's 16 pieces (1 KiB per wave each) go into each group, in the order the
// groups follow the barrier: 5, 6, 7, then 0, 1, 2, 3, 4 of the next tile.  0 = round 3/4 (all sixteen squeezed into groups 5..7: two
// pieces per MFMA gap - an LDS-DMA piece costs its wave 60..185 cycles of issue and a 16x16x128 MFMA covers 32); the others spread them.
#ifndef F8_DMA_SCHED
#define F8_DMA_SCHED 1
#endif
// Diagnostic builds (tools/gemm_mxfp8_ab.py; results are garbage, only the time means something) - one ingredient of the main loop compiled out:
// 1 no LDS-DMA in the loop, 2 every piece re-reads K-tile 0 (all L2 hits), 3 no fragment reads, 4 no barrier / vmcnt wait in the loop, 5 no epilogue,
// 6 / 7 the first round's workgroups skip part of their K range (desynchronises the later rounds: what a staggered start would buy),
// 8 / 9 / 10 (FFN-up form) the epilogue without its GELU + block maximum / without its global stores / without its LDS staging and barriers
#ifndef F8_ABLATE
#define F8_ABLATE 0
#endif
constexpr int kDmaSched[10][8] = {{6, 5, 5, 0, 0, 0, 0, 0}, {3, 3, 3, 3, 2, 2, 0, 0}, {4, 4, 4, 4, 0, 0, 0, 0}, {2, 2, 2, 3, 3, 2, 2, 0}, {2, 2, 2, 2, 2, 2, 2, 2},
                                  {4, 3, 3, 3, 3, 0, 0, 0}, {3, 3, 3, 3, 3, 1, 0, 0}, {2, 3, 3, 4, 4, 0, 0, 0}, {3, 3, 2, 3, 3, 2, 0, 0}, {4, 4, 4, 2, 2, 0, 0, 0}};
constexpr int dma_sched_n(int s, int g) { return kDmaSched[s][(g + 3) & 7]; }
constexpr int dma_sched_first(int s, int g) {
  int n = 0;
  for (int i = 0; i < ((g + 3) & 7); ++i) n += kDmaSched[s][i];
  return n;
}

#define EPI_BIAS_GELU_Q 7  // (this file only) bias + tanh GELU, then MX-quantised: the next GEMM's A operand instead of a bf16 matrix
#define X8_PIN() __builtin_amdgcn_sched_barrier(0)
#define X8_BAR() __builtin_amdgcn_s_barrier()

// acc (C^T fragment, accumulator file) += W fragment x A fragment; the W fragment's scale is byte SW of sw_, the A fragment's byte SA of sa_
// (byte index = op_sel + 2 op_sel_hi; first operand slot = W)
#define CE_MX_ASM(OS, OH) \
  asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel:" OS " op_sel_hi:" OH : "+a"(acc) : "v"(w), "v"(a), "v"(sw_), "v"(sa_))
template <int SW, int SA>
__device__ __forceinline__ void mma_mx(f32x4& acc, const i32x8& w, const i32x8& a, uint32_t sw_, uint32_t sa_) {
  constexpr int lo = (SW & 1) | ((SA & 1) << 1), hi = (SW >> 1) | ((SA >> 1) << 1);
  if constexpr (lo == 0 && hi == 0) CE_MX_ASM("[0,0,0]", "[0,0,0]");
  else if constexpr (lo == 1 && hi == 0) CE_MX_ASM("[1,0,0]", "[0,0,0]");
  else if constexpr (lo == 2 && hi == 0) CE_MX_ASM("[0,1,0]", "[0,0,0]");
  else if constexpr (lo == 3 && hi == 0) CE_MX_ASM("[1,1,0]", "[0,0,0]");
  else if constexpr (lo == 0 && hi == 1) CE_MX_ASM("[0,0,0]", "[1,0,0]");
  else if constexpr (lo == 1 && hi == 1) CE_MX_ASM("[1,0,0]", "[1,0,0]");
  else if constexpr (lo == 2 && hi == 1) CE_MX_ASM("[0,1,0]", "[1,0,0]");
  else if constexpr (lo == 3 && hi == 1) CE_MX_ASM("[1,1,0]", "[1,0,0]");
  else if constexpr (lo == 0 && hi == 2) CE_MX_ASM("[0,0,0]", "[0,1,0]");
  else if constexpr (lo == 1 && hi == 2) CE_MX_ASM("[1,0,0]", "[0,1,0]");
  else if constexpr (lo == 2 && hi == 2) CE_MX_ASM("[0,1,0]", "[0,1,0]");
  else if constexpr (lo == 3 && hi == 2) CE_MX_ASM("[1,1,0]", "[0,1,0]");
  else if constexpr (lo == 0 && hi == 3) CE_MX_ASM("[0,0,0]", "[1,1,0]");
  else if constexpr (lo == 1 && hi == 3) CE_MX_ASM("[1,0,0]", "[1,1,0]");
  else if constexpr (lo == 2 && hi == 3) CE_MX_ASM("[0,1,0]", "[1,1,0]");
  else CE_MX_ASM("[1,1,0]", "[1,1,0]");
}
#undef CE_MX_ASM

template <int EPI, bool MX = false, bool GP = true>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_fp8_w4(
    const unsigned char* __restrict__ A, const unsigned char* __restrict__ W, bf16* __restrict__ C, const float* __restrict__ sa,
    const float* __restrict__ sw, const float* __restrict__ bias, const float* __restrict__ gate, const bf16* __restrict__ res, int M,
    int N, int K, int lda, int ldw, int ldc, int ldres, int gate_rows, int tiles_m, int tiles_n, int t_full, int split,
    float* __restrict__ ws, unsigned char* __restrict__ qs_out) {
  // (MX: sa / sw point at the tiled E8M0 scale bytes.  EPI_BIAS_GELU_Q: C is the e4m3 output [M][ldc BYTES], qs_out its tiled scales)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fg = lane >> 4;

  // split-K tail (as in ce_gemm256w4.hip): the tiles of a partially filled last round are cut along K into `split` pieces that write
  // SCALED fp32 slabs (sa[m] sw[n] acc: the sum of the pieces is then the scaled sum) for gemm256w4_reduce, which adds the bias and
  // applies the epilogue
  const bool partial = (int)blockIdx.x >= t_full;
  int wg, kt0 = 0, ktn = K / BKB;
  if (!partial) {
    wg = xcd_remap(blockIdx.x, t_full);
  } else {
    const int tb = blockIdx.x - t_full;
    wg = t_full + tb / split;
    ktn = ktn / split;
    kt0 = (tb % split) * ktn;
  }
  int m0, n0;
  {
    constexpr int GROUP = 4;
    const int group_sz = GROUP * tiles_n, gid = wg / group_sz, first_m = gid * GROUP;
    const int gm = min(tiles_m - first_m, GROUP);
    m0 = (first_m + (wg % group_sz) % gm) * BM;
    n0 = ((wg % group_sz) / gm) * BN;
  }
#if F8_ABLATE == 6 || F8_ABLATE == 7
  // desynchronisation probe: the workgroups of the FIRST round skip 0 / 1 / 2 / 3 quarters of their K range (6) or idle-free variant 7: only
  // every second one skips a half - the rounds that follow then end at different times on different CUs (garbage in those tiles: timing only)
  if (!partial && (int)blockIdx.x < 256) {
    const int q = (F8_ABLATE == 6 ? (int)(blockIdx.x >> 3) & 3 : 2 * ((int)(blockIdx.x >> 3) & 1)) * ((ktn / 4) & ~1);
    kt0 += q;
    ktn -= q;
  }
#endif
  const int kt_last = ktn - 1;

  // LDS-DMA sources: piece p of this wave = rows 8 (wave + 4 p) .. + 8 of the operand tile, lane l -> row + (l >> 3), slot l & 7
  uint32_t a_voff[8], w_voff[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int row = 8 * (wave + 4 * p) + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    a_voff[p] = (uint32_t)min(m0 + row, M - 1) * (uint32_t)lda + chunk * 16;
#if F8_EPI_LDS
    const int wrow = row;
#else
    // register-direct epilogue: LDS row 16 G + i of a wave's 128 W rows (fragment G, fragment row i) holds W row 8 i + G, so that lane i of an
    // accumulator quad-row owns the 8 CONSECUTIVE output columns 8 i .. 8 i + 7 over its eight G accumulators (a permutation of whole 128-byte
    // rows on the DMA's source side: free)
    const int wrow = (row & 128) | ((row & 15) << 3) | ((row & 127) >> 4);
#endif
    w_voff[p] = (uint32_t)min(n0 + wrow, N - 1) * (uint32_t)ldw + chunk * 16;
  }
  const auto a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, 0xffffffffu, 0x00020000);
  const auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, 0xffffffffu, 0x00020000);
#if F8_ABLATE == 2
  auto koff = [&](int t) __attribute__((always_inline)) -> int { return (kt0 + min(t, 1)) * BKB; };
#else
  auto koff = [&](int t) __attribute__((always_inline)) -> int { return (kt0 + min(t, kt_last)) * BKB; };
#endif
  // piece q (0..15) of a tile: 0..7 = A, 8..15 = W
  auto dma = [&](int q, int stage_bytes, int soff) __attribute__((always_inline)) {
    if (q < 8)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_void*)(smem + stage_bytes + (wave + 4 * q) * 1024), 16, a_voff[q & 7], soff, 0, 0);
    else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void*)(smem + stage_bytes + TILE + (wave + 4 * (q - 8)) * 1024), 16,
                                               w_voff[q & 7], soff, 0, 0);
  };

  // fragment = the 32 bytes k = 32 fg + [0, 32) of a row: chunks 2 fg and 2 fg + 1, in slots (2 fg + h) ^ ((row >> 1) & 7) = (2 fg + h) ^ (fr >> 1)
  int a_rd[2], w_rd[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int chunk = MX ? fg + 4 * h : 2 * fg + h;  // (MX: the hardware's block geometry, see the header)
    a_rd[h] = (wm * 128 + fr) * 128 + ((chunk ^ (fr >> 1)) << 4);
    w_rd[h] = TILE + (wn * 128 + fr) * 128 + ((chunk ^ (fr >> 1)) << 4);
  }
  auto read_frag_real = [&](int base0, int base1, int stage_bytes, int f) __attribute__((always_inline)) -> i32x8 {
    const u32x4 lo = *reinterpret_cast<const u32x4*>(smem + base0 + stage_bytes + f * 2048);
    const u32x4 hi = *reinterpret_cast<const u32x4*>(smem + base1 + stage_bytes + f * 2048);
    return i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
  };

#if F8_ABLATE == 3
  i32x8 ablate_frag = read_frag_real(a_rd[0], a_rd[1], 0, 0);
  bool in_loop = false;
  auto read_frag = [&](int base0, int base1, int stage_bytes, int f) __attribute__((always_inline)) -> i32x8 {
    if (in_loop) {
      asm volatile("" : "+v"(ablate_frag));
      return ablate_frag;
    }
    return read_frag_real(base0, base1, stage_bytes, f);
  };
#else
  auto& read_frag = read_frag_real;
#endif
  f32x4 acc[8][8];
#pragma unroll
  for (int f = 0; f < 8; ++f)
#pragma unroll
    for (int g = 0; g < 8; ++g) acc[f][g] = f32x4{0.f, 0.f, 0.f, 0.f};

  // MX: this lane's scale bytes of its 8 A and 8 W fragments for ONE K-tile = 8 + 8 bytes (see the header); two sets by tile parity
  u32x2 sAq[2], sWq[2];
  const int ktiles = K / BKB;
  const unsigned char* sa_base = nullptr;
  const unsigned char* sw_base = nullptr;
  if (MX) {
    const int rba = min(m0 / 128 + wm, (M - 1) / 128), rbw = min(n0 / 128 + wn, (N - 1) / 128);  // (row blocks past the end: scales of rows never stored)
    sa_base = reinterpret_cast<const unsigned char*>(sa) + (size_t)rba * ktiles * 512 + fg * 128 + fr * 8;
    sw_base = reinterpret_cast<const unsigned char*>(sw) + (size_t)rbw * ktiles * 512 + fg * 128 + fr * 8;
  }
  auto load_scales = [&](int t, int par) __attribute__((always_inline)) {
    const int ta = kt0 + min(t, kt_last);  // (surplus prefetch: re-read the last tile)
    // asm loads (hipcc does not count them): they are issued at the top of a K-tile, in FRONT of that tile's LDS-DMA pieces, so the
    // tile's own "vmcnt(0)" in group 5 retires them - and the W4 / X8 waits stay exactly as counted.  As compiler-visible loads hipcc
    // waited for them at the top of the next tile with vmcnt(2): every DMA piece in flight had to land five groups early.
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(sAq[par]) : "v"(sa_base + (size_t)ta * 512));
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(sWq[par]) : "v"(sw_base + (size_t)ta * 512));
  };
  if (MX) load_scales(0, 0);

#if !F8_A3
  // prologue: tiles 0 and 1 on their way, tile 0 landed; its W fragments and its first three A fragments read
  {
    const int k0 = koff(0), k1 = koff(1);
#pragma unroll
    for (int q = 0; q < 16; ++q) dma(q, 0, k0);
#pragma unroll
    for (int q = 0; q < 16; ++q) dma(q, STAGE, k1);
  }
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  if (MX) asm volatile("" : "+v"(sAq[0]), "+v"(sWq[0]));  // (requested in front of the 32 pieces above)
  X8_BAR();
  i32x8 ring[4], bw0[8], bw1[8];
#pragma unroll
  for (int g = 0; g < 8; ++g) bw0[g] = read_frag(w_rd[0], w_rd[1], 0, g);
#pragma unroll
  for (int f = 0; f < 3; ++f) ring[f] = read_frag(a_rd[0], a_rd[1], 0, f);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  X8_PIN();

  // F8_EPI_LDS: W fragment first - the accumulator holds C^T, lane (fr, fg) of acc[F][G] owns row F*16 + fr and the columns G*16 + fg*4 + [0,4)
#define X8_MMA(F, G, BW)                                                                                                          \
  do {                                                                                                                            \
    if (F8_EPI_LDS) {                                                                                                             \
      if (MX) mma_mx<(G) & 3, (F) & 3>(acc[F][G], (BW)[G], ring[(F) % 4], sWq[SEL_][(G) >> 2], sAq[SEL_][(F) >> 2]);                \
      else asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, %0" : "+a"(acc[F][G]) : "v"((BW)[G]), "v"(ring[(F) % 4]));        \
    } else { /* A fragment first: the accumulator holds C, lane (fr, fg) owns rows F*16 + fg*4 + [0,4) and W fragment row fr */     \
      if (MX) mma_mx<(F) & 3, (G) & 3>(acc[F][G], ring[(F) % 4], (BW)[G], sAq[SEL_][(F) >> 2], sWq[SEL_][(G) >> 2]);                \
      else asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, %0" : "+a"(acc[F][G]) : "v"(ring[(F) % 4]), "v"((BW)[G]));        \
    }                                                                                                                             \
  } while (0)
  // group G of the tile in stage PAR (its W fragments in BW, the next tile's go to BN); fillers between single MFMAs.
  // LDS-DMA schedule (F8_DMA_SCHED, table kDmaSched): piece s of the 16 is issued in the group the table gives, one piece per slot, the
  // slots being the gaps behind MFMAs 1, 3, 5, 7 of a group.  Groups 5..7 of tile T carry the first pieces of tile T+2 (-> stage PAR, free
  // since the barrier in group 5), groups 0..4 of the NEXT tile carry the rest (from there: tile T+1 -> the other stage); all of a tile's
  // pieces are in flight before the vmcnt(0) + barrier in group 5 of the tile in front of it.
#define X8_DMA(G, J, PAR, KN2, KN1)                                                                                           \
  if constexpr (F8_ABLATE != 1 && (J) < dma_sched_n(F8_DMA_SCHED, (G))) {                                                                      \
    if constexpr ((G) >= 5) dma(dma_sched_first(F8_DMA_SCHED, (G)) + (J), (PAR) * STAGE, KN2);                                 \
    else dma(dma_sched_first(F8_DMA_SCHED, (G)) + (J), (1 - (PAR)) * STAGE, KN1);                                              \
  }
#define X8_GROUP(G, PAR, BW, BN_, KNEXT, KN1, SEL)                                                                            \
  {                                                                                                                           \
    constexpr int SEL_ = (SEL);                            /* scale register set = tile parity (MX) */                        \
    constexpr int fn_ = ((G) + 3) & 7;                     /* the A fragment fetched now ... */                                \
    constexpr int sn_ = ((G) + 3 >= 8) ? (1 - (PAR)) * STAGE : (PAR) * STAGE; /* ... from this tile or the next */             \
    if ((G) == 5 && F8_ABLATE != 4) {                                                                                         \
      asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");                                                 \
      if (MX) asm volatile("" : "+v"(sAq[1 - SEL_]), "+v"(sWq[1 - SEL_])); /* the next tile's scales have landed too */         \
      X8_BAR();                                                                                                               \
      X8_PIN();                                                                                                               \
    }                                                                                                                         \
    X8_MMA(G, 0, BW);                                                                                                         \
    ring[((G) + 3) % 4] = read_frag(a_rd[0], a_rd[1], sn_, fn_);                                                               \
    X8_PIN();                                                                                                                 \
    X8_MMA(G, 1, BW);                                                                                                         \
    X8_DMA(G, 0, PAR, KNEXT, KN1)                                                                                             \
    X8_PIN();                                                                                                                 \
    X8_MMA(G, 2, BW);                                                                                                         \
    if ((G) >= 5) { (BN_)[3 * ((G) - 5) + 0] = read_frag(w_rd[0], w_rd[1], (1 - (PAR)) * STAGE, 3 * ((G) - 5) + 0); }            \
    X8_PIN();                                                                                                                 \
    X8_MMA(G, 3, BW);                                                                                                         \
    X8_DMA(G, 1, PAR, KNEXT, KN1)                                                                                             \
    if constexpr (dma_sched_n(F8_DMA_SCHED, (G)) > 4) { X8_DMA(G, 4, PAR, KNEXT, KN1) }                                        \
    X8_PIN();                                                                                                                 \
    X8_MMA(G, 4, BW);                                                                                                         \
    if ((G) >= 5) { (BN_)[3 * ((G) - 5) + 1] = read_frag(w_rd[0], w_rd[1], (1 - (PAR)) * STAGE, 3 * ((G) - 5) + 1); }            \
    X8_PIN();                                                                                                                 \
    X8_MMA(G, 5, BW);                                                                                                         \
    X8_DMA(G, 2, PAR, KNEXT, KN1)                                                                                             \
    if constexpr (dma_sched_n(F8_DMA_SCHED, (G)) > 5) { X8_DMA(G, 5, PAR, KNEXT, KN1) }                                        \
    X8_PIN();                                                                                                                 \
    X8_MMA(G, 6, BW);                                                                                                         \
    if ((G) == 5 || (G) == 6) { (BN_)[3 * ((G) - 5) + 2] = read_frag(w_rd[0], w_rd[1], (1 - (PAR)) * STAGE, 3 * ((G) - 5) + 2); } \
    X8_PIN();                                                                                                                 \
    X8_MMA(G, 7, BW);                                                                                                         \
    X8_DMA(G, 3, PAR, KNEXT, KN1)                                                                                             \
    X8_PIN();                                                                                                                 \
  }
  // W fragments: group 5 -> 0,1,2, group 6 -> 3,4,5, group 7 -> 6,7
#define X8_TILE(PAR, T, BW, BN_, SEL)                                                                                         \
  {                                                                                                                           \
    const int knext = koff((T) + 2), kn1 = koff((T) + 1);                                                                     \
    X8_GROUP(0, PAR, BW, BN_, knext, kn1, SEL) X8_GROUP(1, PAR, BW, BN_, knext, kn1, SEL) X8_GROUP(2, PAR, BW, BN_, knext, kn1, SEL) \
    X8_GROUP(3, PAR, BW, BN_, knext, kn1, SEL) X8_GROUP(4, PAR, BW, BN_, knext, kn1, SEL) X8_GROUP(5, PAR, BW, BN_, knext, kn1, SEL) \
    X8_GROUP(6, PAR, BW, BN_, knext, kn1, SEL) X8_GROUP(7, PAR, BW, BN_, knext, kn1, SEL)                                      \
  }
  {
    const int npairs = ktn >> 1;
#if F8_ABLATE == 3
    in_loop = true;
#endif
    for (int it = 0; it < npairs; ++it) {
      const int t = 2 * it;
      if (MX) load_scales(t + 1, 1);  // (one K-tile ahead: ~1.7 us of matrix work between the request and the first MFMA that reads it)
      X8_TILE(0, t, bw0, bw1, 0)
      if (MX) load_scales(t + 2, 0);
      X8_TILE(1, t + 1, bw1, bw0, 1)
    }
  }
#else
  // ---- F8_A3: A ring of three stages, W ring of two ----
  auto dma_a = [&](int q, int a_off, int soff) __attribute__((always_inline)) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_void*)(smem + a_off + (wave + 4 * q) * 1024), 16, a_voff[q & 7], soff, 0, 0);
  };
  auto dma_w = [&](int q, int par, int soff) __attribute__((always_inline)) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void*)(smem + W_RING + par * TILE + (wave + 4 * q) * 1024), 16, w_voff[q & 7], soff, 0, 0);
  };
  auto read_a = [&](const int (&base)[2], int f) __attribute__((always_inline)) -> i32x8 {
    const u32x4 lo = *reinterpret_cast<const u32x4*>(smem + base[0] + f * 2048);
    const u32x4 hi = *reinterpret_cast<const u32x4*>(smem + base[1] + f * 2048);
    return i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
  };
  auto read_w = [&](int par, int f) __attribute__((always_inline)) -> i32x8 {  // (w_rd already carries TILE: the W half of the two-stage layout)
    const u32x4 lo = *reinterpret_cast<const u32x4*>(smem + w_rd[0] - TILE + W_RING + par * TILE + f * 2048);
    const u32x4 hi = *reinterpret_cast<const u32x4*>(smem + w_rd[1] - TILE + W_RING + par * TILE + f * 2048);
    return i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
  };
  // prologue: tiles 0 and 1 on their way (A stages 0, 1; W stages 0, 1), tile 0 landed; its W fragments and its first three A fragments read
  {
    const int k0 = koff(0), k1 = koff(1);
#pragma unroll
    for (int q = 0; q < 8; ++q) dma_a(q, 0, k0);
#pragma unroll
    for (int q = 0; q < 8; ++q) dma_w(q, 0, k0);
#pragma unroll
    for (int q = 0; q < 8; ++q) dma_a(q, TILE, k1);
#pragma unroll
    for (int q = 0; q < 8; ++q) dma_w(q, 1, k1);
  }
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  if (MX) asm volatile("" : "+v"(sAq[0]), "+v"(sWq[0]));
  X8_BAR();
  i32x8 ring[4], bw0[8], bw1[8];
  int arc[2] = {a_rd[0], a_rd[1]}, arn[2] = {a_rd[0] + TILE, a_rd[1] + TILE};  // fragment read bases of the tile being multiplied / of the next one
  int a_c = 0, a_n = TILE, a_nn = 2 * TILE;                                     // A stages (bytes, wave-uniform) of tiles T, T+1 and the one tile T+2 lands in
#pragma unroll
  for (int g = 0; g < 8; ++g) bw0[g] = read_w(0, g);
#pragma unroll
  for (int f = 0; f < 3; ++f) ring[f] = read_a(arc, f);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  X8_PIN();

#define X8_MMA(F, G, BW)                                                                                                          \
  do {                                                                                                                            \
    if (F8_EPI_LDS) {                                                                                                             \
      if (MX) mma_mx<(G) & 3, (F) & 3>(acc[F][G], (BW)[G], ring[(F) % 4], sWq[SEL_][(G) >> 2], sAq[SEL_][(F) >> 2]);                \
      else asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, %0" : "+a"(acc[F][G]) : "v"((BW)[G]), "v"(ring[(F) % 4]));        \
    } else { /* A fragment first: the accumulator holds C, lane (fr, fg) owns rows F*16 + fg*4 + [0,4) and W fragment row fr */     \
      if (MX) mma_mx<(F) & 3, (G) & 3>(acc[F][G], ring[(F) % 4], (BW)[G], sAq[SEL_][(F) >> 2], sWq[SEL_][(G) >> 2]);                \
      else asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, %0" : "+a"(acc[F][G]) : "v"(ring[(F) % 4]), "v"((BW)[G]));        \
    }                                                                                                                             \
  } while (0)
  // the two LDS-DMA pieces of group G (J = 0, 1): group 0 - the last two W pieces of tile T+1; groups 1..4 - the eight A pieces of tile T+2 (-> the
  // stage tile T-1 left at the barrier of T-1); groups 5..7 - the first six W pieces of tile T+2 (-> W stage PAR, free since this tile's barrier)
#define X8_DMA(G, J, PAR, KN2, KN1)                                                                                           \
  if constexpr (F8_ABLATE != 1) {                                                                                             \
    if constexpr ((G) == 0) dma_w(6 + (J), 1 - (PAR), KN1);                                                                   \
    else if constexpr ((G) <= 4) dma_a(2 * ((G) - 1) + (J), a_nn, KN2);                                                        \
    else dma_w(2 * ((G) - 5) + (J), (PAR), KN2);                                                                              \
  }
#define X8_GROUP(G, PAR, BW, BN_, KNEXT, KN1, SEL)                                                                            \
  {                                                                                                                           \
    constexpr int SEL_ = (SEL);                                                                                               \
    constexpr int fn_ = ((G) + 3) & 7;                                                                                        \
    if ((G) == 5 && F8_ABLATE != 4) {                                                                                         \
      /* everything older than the eight A pieces of tile T+2 (groups 1..4) has landed: tile T+1 whole, its scales */          \
      asm volatile("s_waitcnt vmcnt(8)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");                                                 \
      if (MX) asm volatile("" : "+v"(sAq[1 - SEL_]), "+v"(sWq[1 - SEL_]));                                                     \
      X8_BAR();                                                                                                               \
      X8_PIN();                                                                                                               \
    }                                                                                                                         \
    X8_MMA(G, 0, BW);                                                                                                         \
    if constexpr ((G) + 3 >= 8) ring[((G) + 3) % 4] = read_a(arn, fn_); else ring[((G) + 3) % 4] = read_a(arc, fn_);          \
    X8_PIN();                                                                                                                 \
    X8_MMA(G, 1, BW);                                                                                                         \
    X8_DMA(G, 0, PAR, KNEXT, KN1)                                                                                             \
    X8_PIN();                                                                                                                 \
    X8_MMA(G, 2, BW);                                                                                                         \
    if ((G) >= 5) { (BN_)[3 * ((G) - 5) + 0] = read_w(1 - (PAR), 3 * ((G) - 5) + 0); }                                         \
    X8_PIN();                                                                                                                 \
    X8_MMA(G, 3, BW);                                                                                                         \
    X8_PIN();                                                                                                                 \
    X8_MMA(G, 4, BW);                                                                                                         \
    if ((G) >= 5) { (BN_)[3 * ((G) - 5) + 1] = read_w(1 - (PAR), 3 * ((G) - 5) + 1); }                                         \
    X8_PIN();                                                                                                                 \
    X8_MMA(G, 5, BW);                                                                                                         \
    X8_DMA(G, 1, PAR, KNEXT, KN1)                                                                                             \
    X8_PIN();                                                                                                                 \
    X8_MMA(G, 6, BW);                                                                                                         \
    if ((G) == 5 || (G) == 6) { (BN_)[3 * ((G) - 5) + 2] = read_w(1 - (PAR), 3 * ((G) - 5) + 2); }                             \
    X8_PIN();                                                                                                                 \
    X8_MMA(G, 7, BW);                                                                                                         \
    X8_PIN();                                                                                                                 \
  }
#define X8_TILE(PAR, T, BW, BN_, SEL)                                                                                         \
  {                                                                                                                           \
    const int knext = koff((T) + 2), kn1 = koff((T) + 1);                                                                     \
    X8_GROUP(0, PAR, BW, BN_, knext, kn1, SEL) X8_GROUP(1, PAR, BW, BN_, knext, kn1, SEL) X8_GROUP(2, PAR, BW, BN_, knext, kn1, SEL) \
    X8_GROUP(3, PAR, BW, BN_, knext, kn1, SEL) X8_GROUP(4, PAR, BW, BN_, knext, kn1, SEL) X8_GROUP(5, PAR, BW, BN_, knext, kn1, SEL) \
    X8_GROUP(6, PAR, BW, BN_, knext, kn1, SEL) X8_GROUP(7, PAR, BW, BN_, knext, kn1, SEL)                                      \
    /* rotate the A ring: the next tile becomes the current one, tile T+2's stage the next one, this tile's stage takes tile T+3 */ \
    {                                                                                                                         \
      const int freed = a_c;                                                                                                  \
      a_c = a_n; a_n = a_nn; a_nn = freed;                                                                                    \
      arc[0] = arn[0]; arc[1] = arn[1];                                                                                       \
      arn[0] = a_rd[0] + a_n; arn[1] = a_rd[1] + a_n;                                                                         \
    }                                                                                                                         \
  }
  {
    const int npairs = ktn >> 1;
    for (int it = 0; it < npairs; ++it) {
      const int t = 2 * it;
      if (MX) load_scales(t + 1, 1);
      X8_TILE(0, t, bw0, bw1, 0)
      if (MX) load_scales(t + 2, 0);
      X8_TILE(1, t + 1, bw1, bw0, 1)
    }
  }
#endif
#undef X8_TILE
#undef X8_GROUP
#undef X8_DMA
#undef X8_MMA
  asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#if F8_EPI_LDS
  X8_BAR();  // (the staged epilogue reuses the LDS stages)
#endif
#if F8_ABLATE == 5
  {
    float sum = 0.f;
#pragma unroll
    for (int f = 0; f < 8; ++f)
#pragma unroll
      for (int g = 0; g < 8; ++g) sum += acc[f][g][0] + acc[f][g][1] + acc[f][g][2] + acc[f][g][3];
    if (sum == 123.456f) ws[0] = sum;
    return;
  }
#endif

#if F8_EPI_LDS
  // ---- epilogue: scales, bias -> bf16 -> LDS, four passes of 64 staged rows (pass p: accumulator rows f = 2p, 2p+1 of every wave),
  // then the row-contiguous half of ce_gemm_epi.h (activation / gated residual, 16-byte stores)
  f32x4 swv[8], bvv[8];
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const int nc = min(n0 + wn * 128 + g * 16 + fg * 4, N - 4);
    swv[g] = MX ? f32x4{1.f, 1.f, 1.f, 1.f} : *reinterpret_cast<const f32x4*>(sw + nc);  // (MX: the matrix pipe applied the scales)
    bvv[g] = bias != nullptr ? *reinterpret_cast<const f32x4*>(bias + nc) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  if (partial) {  // fp32 slab [wave][f][g][lane], already scaled
    float* slab = ws + (size_t)(blockIdx.x - t_full) * (BM * BN);
#pragma unroll
    for (int f = 0; f < 8; ++f) {
      const float sav = MX ? 1.0f : sa[min(m0 + wm * 128 + f * 16 + fr, M - 1)];
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const f32x4 v = acc[f][g];
        const f32x4 o = {v[0] * (sav * swv[g][0]), v[1] * (sav * swv[g][1]), v[2] * (sav * swv[g][2]), v[3] * (sav * swv[g][3])};
        *reinterpret_cast<f32x4*>(slab + (((wave * 64 + f * 8 + g) * 64) + lane) * 4) = o;
      }
    }
    return;
  }
  // Gated residual: all of this thread's residual chunks (4 passes x 8 x 16 B) and its gate values are requested HERE, before the first
  // staging pass, and the stores are predicated by a buffer descriptor's range check (as in ce_gemm256w4.hip: per-pass loads cost four
  // serial memory round trips per tile).  The launcher sends gate rows shorter than a tile to the 8-wave kernel.
  // (GP = false - gate rows shorter than a tile, or a C beyond 32-bit byte offsets: the per-pass path of ce_gemm_epi.h)
  constexpr bool prefetch = EPI == EPI_GATE_RES && GP;
  u32x4 rv[4][8];
  f32x4 gA0, gA1, gB0, gB1;
  int g_switch = 0x7fffffff;
  const int my_n = n0 + (tid & 31) * 8, my_nc = min(my_n, N - 8);
  const auto c_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)C, 0, (uint32_t)(M - 1) * (uint32_t)(ldc * 2) + (uint32_t)N * 2u, 0x00020000);
  if (prefetch) {
    gA0 = gA1 = gB0 = gB1 = f32x4{1.f, 1.f, 1.f, 1.f};
    if (gate != nullptr) {
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
      for (int tt = 0; tt < 8; ++tt) {
        const int rl = (tid + 256 * tt) >> 5;
        const int m = min(m0 + (rl >> 5) * 128 + p * 32 + (rl & 31), M - 1);
        rv[p][tt] = *reinterpret_cast<const u32x4*>(res + (size_t)m * ldres + my_nc);
      }
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    if (p > 0 && F8_ABLATE != 10) __syncthreads();
#pragma unroll
    for (int ff = 0; ff < (F8_ABLATE == 10 ? 0 : 2); ++ff) {  /* (10: no staging writes, no barriers: the chunk phase reads stale LDS) */
      const int f = 2 * p + ff;
      const float sav = MX ? 1.0f : sa[min(m0 + wm * 128 + f * 16 + fr, M - 1)];
      const int rl = wm * 32 + ff * 16 + fr;
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const int cl = wn * 128 + g * 16 + fg * 4;
        const f32x4 v = acc[f][g];
        const u32x2 pk = {pack_bf16(v[0] * (sav * swv[g][0]) + bvv[g][0], v[1] * (sav * swv[g][1]) + bvv[g][1]),
                          pack_bf16(v[2] * (sav * swv[g][2]) + bvv[g][2], v[3] * (sav * swv[g][3]) + bvv[g][3])};
        *reinterpret_cast<u32x2*>(smem + rl * CROW + cl * 2) = pk;
      }
    }
    if (F8_ABLATE != 10) __syncthreads();
    if (prefetch) {
#pragma unroll
      for (int tt = 0; tt < 8; ++tt) {
        const int rl = (tid + 256 * tt) >> 5;
        const int m = m0 + (rl >> 5) * 128 + p * 32 + (rl & 31);
        const u32x4 y = *reinterpret_cast<const u32x4*>(smem + rl * CROW + (tid & 31) * 16);
        const bool second = m >= g_switch;
        const f32x4 g0 = second ? gB0 : gA0, g1 = second ? gB1 : gA1;
        const u32x4 r = rv[p][tt];
        u32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float ga = q < 2 ? g0[2 * q] : g1[2 * q - 4], gb = q < 2 ? g0[2 * q + 1] : g1[2 * q - 3];
          o[q] = pack_bf16(mul_then_add(bf16lo(y[q]), ga, bf16lo(r[q])), mul_then_add(bf16hi(y[q]), gb, bf16hi(r[q])));
        }
        const uint32_t coff = (m < M && my_n < N) ? (uint32_t)m * (uint32_t)(ldc * 2) + (uint32_t)my_n * 2u : 0xffffffffu;
        __builtin_amdgcn_raw_buffer_store_b128(o, c_rsrc, coff, 0, 0);
      }
    } else if (EPI == EPI_BIAS_GELU_Q) {
      // GELU on the staged bf16 rows, then ce_quant_rows_mxfp8's contract on the bf16 result: chunk cc (8 columns) of a staged row sits with
      // its block mates cc ^ 1, cc ^ 2, cc ^ 3 in adjacent lanes (c = tid + 256 tt: cc = tid & 31), so the block amax is two lane exchanges
      unsigned char* q_out = reinterpret_cast<unsigned char*>(C);
#pragma unroll
      for (int tt = 0; tt < 8; ++tt) {
        const int c = tid + 256 * tt;
        const int rl = c >> 5, cc = c & 31;
        const int m = m0 + (rl >> 5) * 128 + p * 32 + (rl & 31), n = n0 + cc * 8;
        const u32x4 y = *reinterpret_cast<const u32x4*>(smem + rl * CROW + cc * 16);
        u32x4 o;
        float am = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#if F8_ABLATE == 8  /* (epilogue ablation: no GELU, no block maximum) */
          o[q] = y[q];
          am = 1.0f;
        }
#else
          o[q] = pack_bf16(gelu_tanh(bf16lo(y[q])), gelu_tanh(bf16hi(y[q])));
          am = fmaxf(am, fmaxf(fabsf(bf16lo(o[q])), fabsf(bf16hi(o[q]))));
        }
        am = fmaxf(am, __shfl_xor(am, 1, 64));
        am = fmaxf(am, __shfl_xor(am, 2, 64));
#endif
        const int byte = mx_scale_byte_nosat(am);
        const float inv = mx_inv_scale(byte);
        int w0 = 0, w1 = 0;
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(bf16lo(o[0]) * inv), clamp448(bf16hi(o[0]) * inv), w0, false);
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(bf16lo(o[1]) * inv), clamp448(bf16hi(o[1]) * inv), w0, true);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(bf16lo(o[2]) * inv), clamp448(bf16hi(o[2]) * inv), w1, false);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(bf16lo(o[3]) * inv), clamp448(bf16hi(o[3]) * inv), w1, true);
        if (m < M && n < N && (F8_ABLATE != 9 || w0 == 0x12345677)) {  /* (9: no stores) */
          *reinterpret_cast<u32x2*>(q_out + (size_t)m * ldc + n) = u32x2{(uint32_t)w0, (uint32_t)w1};
          if ((cc & 3) == 0) qs_out[mx_gemm_scale_offset(m, n >> 5, N >> 7)] = (unsigned char)byte;
        }
      }
    } else {
      epi_chunks<EPI == EPI_BIAS_GELU_Q ? EPI_BIAS : EPI, 8>(smem, CROW,
                         [&](int tt, int& rl, int& cc, int& mr) {
                           const int c = tid + 256 * tt;
                           rl = c >> 5;
                           cc = c & 31;
                           mr = (rl >> 5) * 128 + p * 32 + (rl & 31);
                         },
                         m0, n0, C, gate, res, M, N, ldc, ldres, gate_rows);
    }
  }

#else
  // ---- epilogue, REGISTER-DIRECT (round 6).  With the A fragment as the FIRST operand the accumulator holds C: lane (fr, fg) of acc[F][G]
  // owns rows F*16 + fg*4 + [0,4) of the wave tile and the output column of W fragment row fr - and the DMA's source-side row permutation
  // (LDS row 16 G + i <- W row 8 i + G) makes that column 8 fr + G.  So for a fixed row (F, jj) a lane's eight G accumulators are EIGHT
  // CONSECUTIVE COLUMNS: one 16-byte bf16 (8-byte e4m3) store per lane, the sixteen fr lanes of a quad-row cover 256 (128) contiguous bytes of
  // the row - the row-contiguous stores the LDS staging existed for, without the LDS round trip, its eight barriers and the 24 KiB of LDS
  // traffic per pass; an MX block of 32 output columns is the four lanes of a DPP quad.  Same arithmetic, same roundings, same order as the
  // staged form (tests: bit-identical to ce_gemm_mxfp8 / ce_quant_rows_mxfp8 compositions and to the F8_EPI_LDS build).
  const int col0 = n0 + wn * 128 + fr * 8;  // this lane's 8 output columns
  const bool col_ok = col0 < N;
  const int colc = min(col0, N - 8);
  const int row_base = m0 + wm * 128 + fg * 4;  // + F*16 + jj
  f32x4 swv[2], bvv[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    swv[h] = MX ? f32x4{1.f, 1.f, 1.f, 1.f} : *reinterpret_cast<const f32x4*>(sw + colc + 4 * h);  // (MX: the matrix pipe applied the scales)
    bvv[h] = bias != nullptr ? *reinterpret_cast<const f32x4*>(bias + colc + 4 * h) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  if (partial) {
    // fp32 slab, already scaled, in the [wave][f][g][lane'] order gemm256w4_reduce / gemm_fp8w4_reduce_gelu_q read (the C^T accumulator order of
    // the staged form: lane' (fr', fg') of [f][g] = row f*16 + fr', columns g*16 + fg'*4 + [0,4)): this lane's (F, jj, G = 4 h .. 4 h + 3) is row
    // F*16 + fg*4 + jj, columns fr*8 + 4 h + [0,4) -> f = F, fr' = fg*4 + jj, g = fr >> 1, fg' = 2 (fr & 1) + h
    float* slab = ws + (size_t)(blockIdx.x - t_full) * (BM * BN);
#pragma unroll
    for (int F = 0; F < 8; ++F) {
      f32x4 av[8];
#pragma unroll
      for (int G = 0; G < 8; ++G) {
        asm volatile("" : "+a"(acc[F][G]));  // (pins the read-out of fragment row F here, as in the main path below)
        av[G] = acc[F][G];
      }
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const float sav = MX ? 1.0f : sa[min(row_base + F * 16 + jj, M - 1)];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x4 o = {av[4 * h + 0][jj] * (sav * swv[h][0]), av[4 * h + 1][jj] * (sav * swv[h][1]),
                           av[4 * h + 2][jj] * (sav * swv[h][2]), av[4 * h + 3][jj] * (sav * swv[h][3])};
          const int lane_o = fg * 4 + jj + 16 * (2 * (fr & 1) + h);
          *reinterpret_cast<f32x4*>(slab + (((wave * 64 + F * 8 + (fr >> 1)) * 64) + lane_o) * 4) = o;
        }
      }
    }
    return;
  }
  // Gated residual: this lane's 32 residual chunks (8 F x 4 rows x 16 B) and its gate values are requested HERE, in front of the first
  // row (the main loop's fragment registers are dead); stores are predicated by the buffer descriptor's range check.
  // (GP = false - gate rows shorter than a tile, or a C beyond 32-bit byte offsets: per-row loads and 64-bit addresses)
  constexpr bool prefetch = EPI == EPI_GATE_RES && GP;
  u32x4 rv[8][4];
  f32x4 gA[2], gB[2];
  int g_switch = 0x7fffffff;
  const auto c_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)C, 0, (uint32_t)(M - 1) * (uint32_t)(ldc * (EPI == EPI_BIAS_GELU_Q ? 1 : 2)) +
                                                                        (uint32_t)N * (EPI == EPI_BIAS_GELU_Q ? 1u : 2u), 0x00020000);
  if (prefetch) {
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
    for (int F = 0; F < 8; ++F)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
        rv[F][jj] = *reinterpret_cast<const u32x4*>(res + (size_t)min(row_base + F * 16 + jj, M - 1) * ldres + colc);
  }
  f32x4 sarow[8];  // (per-row-scale form: this lane's 32 row scales, requested together - not beside the gated residual's 128 prefetch registers)
  constexpr bool sa_ahead = !MX && EPI != EPI_GATE_RES;
  if (sa_ahead) {
#pragma unroll
    for (int F = 0; F < 8; ++F)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) sarow[F][jj] = sa[min(row_base + F * 16 + jj, M - 1)];
  }
#pragma unroll
  for (int F = 0; F < 8; ++F) {
    f32x4 av[8];  // the eight accumulators of this fragment row out of the accumulator file, whole
#pragma unroll
    for (int G = 0; G < 8; ++G) {
      asm volatile("" : "+a"(acc[F][G]));  // (pins the read-out of fragment row F here: nothing of it is hoisted above the rows in front of it)
      av[G] = acc[F][G];
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int m = row_base + F * 16 + jj;
      const float sav = MX ? 1.0f : sa_ahead ? sarow[F][jj] : sa[min(m, M - 1)];
      u32x4 y;  // bf16(acc * scales + bias), 8 columns
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int h = q >> 1, e = 2 * (q & 1);
        y[q] = pack_bf16(av[2 * q][jj] * (sav * swv[h][e]) + bvv[h][e], av[2 * q + 1][jj] * (sav * swv[h][e + 1]) + bvv[h][e + 1]);
      }
      const bool ok = m < M && col_ok;
      if (EPI == EPI_BIAS_GELU_Q) {
        // GELU on the bf16 row piece, then ce_quant_rows_mxfp8's contract on the bf16 result: the block's other three 8-column pieces sit in
        // the lanes fr ^ 1, fr ^ 2, fr ^ 3 of the same quad
        unsigned char* q_out = reinterpret_cast<unsigned char*>(C);
        u32x4 o;
        float am = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          o[q] = pack_bf16(gelu_tanh(bf16lo(y[q])), gelu_tanh(bf16hi(y[q])));
          am = fmaxf(am, fmaxf(fabsf(bf16lo(o[q])), fabsf(bf16hi(o[q]))));
        }
        am = fmaxf(am, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(am), 0xB1, 0xf, 0xf, true)));  // quad_perm [1,0,3,2]
        am = fmaxf(am, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(am), 0x4E, 0xf, 0xf, true)));  // quad_perm [2,3,0,1]
        const int byte = mx_scale_byte_nosat(am);
        const float inv = mx_inv_scale(byte);
        int w0 = 0, w1 = 0;
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(bf16lo(o[0]) * inv), clamp448(bf16hi(o[0]) * inv), w0, false);
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(bf16lo(o[1]) * inv), clamp448(bf16hi(o[1]) * inv), w0, true);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(bf16lo(o[2]) * inv), clamp448(bf16hi(o[2]) * inv), w1, false);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(bf16lo(o[3]) * inv), clamp448(bf16hi(o[3]) * inv), w1, true);
        if (GP) {
          const uint32_t qoff = ok ? (uint32_t)m * (uint32_t)ldc + (uint32_t)col0 : 0xffffffffu;
          __builtin_amdgcn_raw_buffer_store_b64(u32x2{(uint32_t)w0, (uint32_t)w1}, c_rsrc, qoff, 0, 0);
        } else if (ok) {
          *reinterpret_cast<u32x2*>(q_out + (size_t)m * ldc + col0) = u32x2{(uint32_t)w0, (uint32_t)w1};
        }
        if (ok && (fr & 3) == 0) qs_out[mx_gemm_scale_offset(m, col0 >> 5, N >> 7)] = (unsigned char)byte;
      } else {
        u32x4 o = y;
        if (EPI == EPI_BIAS_GELU) {
#pragma unroll
          for (int q = 0; q < 4; ++q) o[q] = pack_bf16(gelu_tanh(bf16lo(y[q])), gelu_tanh(bf16hi(y[q])));
        } else if (EPI == EPI_GATE_RES) {
          u32x4 r;
          f32x4 g0, g1;
          if (prefetch) {
            r = rv[F][jj];
            const bool second = m >= g_switch;
            g0 = second ? gB[0] : gA[0];
            g1 = second ? gB[1] : gA[1];
          } else {
            const int mc = min(m, M - 1);
            r = *reinterpret_cast<const u32x4*>(res + (size_t)mc * ldres + colc);
            g0 = g1 = f32x4{1.f, 1.f, 1.f, 1.f};
            if (gate != nullptr) {
              const float* gp = gate + (gate_rows > 0 ? (size_t)(mc / gate_rows) * N : 0) + colc;
              g0 = *reinterpret_cast<const f32x4*>(gp);
              g1 = *reinterpret_cast<const f32x4*>(gp + 4);
            }
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float ga = q < 2 ? g0[2 * q] : g1[2 * q - 4], gb = q < 2 ? g0[2 * q + 1] : g1[2 * q - 3];
            // x.float() + y * gate with both fp32 roundings of the reference (transformer_chronoedit.py:281,293): no fma contraction
            o[q] = pack_bf16(mul_then_add(bf16lo(y[q]), ga, bf16lo(r[q])), mul_then_add(bf16hi(y[q]), gb, bf16hi(r[q])));
          }
        }
        if (GP) {
          const uint32_t coff = ok ? (uint32_t)m * (uint32_t)(ldc * 2) + (uint32_t)col0 * 2u : 0xffffffffu;
          __builtin_amdgcn_raw_buffer_store_b128(o, c_rsrc, coff, 0, 0);
        } else if (ok) {
          *reinterpret_cast<u32x4*>(C + (size_t)m * ldc + col0) = o;
        }
      }
    }
  }
#endif
}

// Split-K tail of the FFN-up form: sums the `split` fp32 slabs of one quadrant (= one producer wave's 128 x 128 accumulators, layout
// [wave][f][g][lane] as the partial tiles above write them) of a tail tile, adds the bias, applies GELU and the MX quantisation of
// EPI_BIAS_GELU_Q.  grid = 4 x the number of tail tiles, 256 threads: thread (w, lane) takes accumulator rows f = 2w, 2w+1 of the quadrant.
__global__ __launch_bounds__(256) void gemm_fp8w4_reduce_gelu_q(unsigned char* __restrict__ q_out, unsigned char* __restrict__ qs_out,
                                                                const float* __restrict__ bias, int M, int N, int ldq, int tiles_m, int tiles_n,
                                                                int t_full, int split, const float* __restrict__ ws) {
  constexpr int QROW = 128 * 2 + 16;
  __shared__ __attribute__((aligned(16))) unsigned char smem[128 * QROW];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int tile = blockIdx.x >> 2, q = blockIdx.x & 3;  // q = producer wave = (wm, wn)
  int m0, n0;
  {
    constexpr int GROUP = 4;  // the raster of gemm_fp8_w4
    const int wg = t_full + tile;
    const int group_sz = GROUP * tiles_n, gid = wg / group_sz, first_m = gid * GROUP;
    const int gm = min(tiles_m - first_m, GROUP);
    m0 = (first_m + (wg % group_sz) % gm) * BM + (q >> 1) * 128;
    n0 = ((wg % group_sz) / gm) * BN + (q & 1) * 128;
  }
  const float* slab = ws + (size_t)tile * split * (BM * BN);
#pragma unroll
  for (int ff = 0; ff < 2; ++ff) {
    const int f = 2 * w + ff;
    const int rl = f * 16 + fr;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const int cl = g * 16 + fg * 4;
      const f32x4 bv = bias != nullptr ? *reinterpret_cast<const f32x4*>(bias + min(n0 + cl, N - 4)) : f32x4{0.f, 0.f, 0.f, 0.f};
      const int e = (((q * 64 + f * 8 + g) * 64) + lane) * 4;
      f32x4 v = *reinterpret_cast<const f32x4*>(slab + e);
      for (int sidx = 1; sidx < split; ++sidx) v += *reinterpret_cast<const f32x4*>(slab + (size_t)sidx * (BM * BN) + e);
      const u32x2 pk = {pack_bf16(v[0] + bv[0], v[1] + bv[1]), pack_bf16(v[2] + bv[2], v[3] + bv[3])};
      *reinterpret_cast<u32x2*>(smem + rl * QROW + cl * 2) = pk;
    }
  }
  __syncthreads();
#pragma unroll
  for (int tt = 0; tt < 8; ++tt) {
    const int c = tid + 256 * tt;
    const int rl = c >> 4, cc = c & 15;
    const int m = m0 + rl, n = n0 + cc * 8;
    const u32x4 y = *reinterpret_cast<const u32x4*>(smem + rl * QROW + cc * 16);
    u32x4 o;
    float am = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      o[k] = pack_bf16(gelu_tanh(bf16lo(y[k])), gelu_tanh(bf16hi(y[k])));
      am = fmaxf(am, fmaxf(fabsf(bf16lo(o[k])), fabsf(bf16hi(o[k]))));
    }
    am = fmaxf(am, __shfl_xor(am, 1, 64));
    am = fmaxf(am, __shfl_xor(am, 2, 64));
    const int byte = mx_scale_byte_nosat(am);
    const float inv = mx_inv_scale(byte);
    int w0 = 0, w1 = 0;
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(bf16lo(o[0]) * inv), clamp448(bf16hi(o[0]) * inv), w0, false);
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(bf16lo(o[1]) * inv), clamp448(bf16hi(o[1]) * inv), w0, true);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(bf16lo(o[2]) * inv), clamp448(bf16hi(o[2]) * inv), w1, false);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(bf16lo(o[3]) * inv), clamp448(bf16hi(o[3]) * inv), w1, true);
    if (m < M && n < N) {
      *reinterpret_cast<u32x2*>(q_out + (size_t)m * ldq + n) = u32x2{(uint32_t)w0, (uint32_t)w1};
      if ((cc & 3) == 0) qs_out[mx_gemm_scale_offset(m, n >> 5, N >> 7)] = (unsigned char)byte;
    }
  }
}

}  // namespace

extern "C" void ce_gemm256_workspace(hipStream_t stream, float** ws, size_t* bytes, int* cus);
extern "C" int ce_gemm256w4_reduce_launch(int epilogue, void* C, const float* bias, const float* gate, const void* res, int M, int N, int ldc,
                                          int ldres, int gate_rows, int tiles_m, int tiles_n, int t_full, int split, const float* ws, int tail,
                                          hipStream_t stream);

static int fp8w4_launch(bool mx, const void* Aq, const void* Wq, void* C, const float* sa, const float* sw, const float* bias,
                        int epilogue, const float* gate, const void* res, int M, int N, int K, int lda, int ldw, int ldc,
                        int ldres, int gate_rows, hipStream_t stream) {
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const int nwg = tiles_m * tiles_n, kt = K / BKB;
  // (the prefetched gated-residual epilogue holds ONE or TWO samples' gate rows per tile and stores through 32-bit offsets)
#if F8_EPI_LDS
  const bool gate_prefetch = epilogue != EPI_GATE_RES || ((gate == nullptr || gate_rows <= 0 || gate_rows >= BM) && (long long)M * ldc * 2 < (1ll << 32));
#else
  // (register-direct epilogue: every form stores through a buffer descriptor's 32-bit offsets when C fits them)
  const bool gate_prefetch = (long long)M * ldc * 2 < (1ll << 32) && (epilogue != EPI_GATE_RES || gate == nullptr || gate_rows <= 0 || gate_rows >= BM);
#endif
  float* g_ws = nullptr;
  size_t g_ws_bytes = 0;
  int g_cus = 256;
  ce_gemm256_workspace(stream, &g_ws, &g_ws_bytes, &g_cus);
  int tail = nwg % g_cus, split = 1;
  if (tail > 0 && g_ws != nullptr) {
    for (int sp = std::min(g_cus / tail, 8); sp >= 2; --sp)
      if (kt % (2 * sp) == 0 && (size_t)tail * sp * BM * BN * sizeof(float) <= g_ws_bytes) {
        split = sp;
        break;
      }
    // An fp8 round is short (1.7 us per K-tile): cutting the last round only pays when the time it saves clearly exceeds what the
    // slabs cost (256 KiB written and read back per piece at ~4 TB/s, plus the reduce launch).  Measured: 116 tail tiles of a
    // K = 5120 product cut in two were 1.5 % SLOWER than run whole; 6 tiles cut in eight, or K = 13824, gain 3 %.
    if (split > 1) {
      const double saving_us = (1.0 - 1.0 / split) * kt * 1.7, slabs_us = (double)tail * split * 0.128 + 8.0;
      if (saving_us < 1.5 * slabs_us) split = 1;
    }
  }
  if (split == 1) tail = 0;
  const int t_full = nwg - tail;
  dim3 grid(t_full + tail * split), block(256);
  const int lds = LDS_BYTES;
  static bool attr_done_[CE_MAX_DEVICES][3] = {};
  bool* attr_done = attr_done_[ce_device_slot()];
#define F8_LAUNCH(E)                                                                                                          \
  do {                                                                                                                        \
    if (!attr_done[E]) {                                                                                                      \
      if (hipFuncSetAttribute((const void*)gemm_fp8_w4<E>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return CE_ERR_ARG; \
      if (hipFuncSetAttribute((const void*)gemm_fp8_w4<E, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return CE_ERR_ARG; \
      if (hipFuncSetAttribute((const void*)gemm_fp8_w4<E, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return CE_ERR_ARG; \
      attr_done[E] = true;                                                                                                    \
    }                                                                                                                         \
    if (mx && !gate_prefetch)                                                                                                 \
      hipLaunchKernelGGL((gemm_fp8_w4<E, true, false>), grid, block, lds, stream, (const unsigned char*)Aq, (const unsigned char*)Wq, (bf16*)C, sa, sw, \
                         bias, gate, (const bf16*)res, M, N, K, lda, ldw, ldc, ldres, gate_rows, tiles_m, tiles_n, t_full, split, g_ws, nullptr); \
    else if (mx)                                                                                                              \
      hipLaunchKernelGGL((gemm_fp8_w4<E, true>), grid, block, lds, stream, (const unsigned char*)Aq, (const unsigned char*)Wq, (bf16*)C, sa, sw, \
                         bias, gate, (const bf16*)res, M, N, K, lda, ldw, ldc, ldres, gate_rows, tiles_m, tiles_n, t_full, split, g_ws, nullptr); \
    else                                                                                                                      \
      hipLaunchKernelGGL((gemm_fp8_w4<E>), grid, block, lds, stream, (const unsigned char*)Aq, (const unsigned char*)Wq, (bf16*)C, sa, sw, \
                         bias, gate, (const bf16*)res, M, N, K, lda, ldw, ldc, ldres, gate_rows, tiles_m, tiles_n, t_full, split, g_ws, nullptr); \
  } while (0)
  switch (epilogue) {
    case EPI_BIAS: F8_LAUNCH(EPI_BIAS); break;
    case EPI_BIAS_GELU: F8_LAUNCH(EPI_BIAS_GELU); break;
    case EPI_GATE_RES: F8_LAUNCH(EPI_GATE_RES); break;
    default: return CE_ERR_ARG;
  }
#undef F8_LAUNCH
  if (tail) {
    const int rc = ce_gemm256w4_reduce_launch(epilogue, C, bias, gate, res, M, N, ldc, ldres, gate_rows, tiles_m, tiles_n, t_full, split, g_ws,
                                              tail, stream);
    if (rc != CE_OK) return rc;
  }
  return (int)hipGetLastError();
}

extern "C" int ce_gemm_fp8w4_launch(const void* Aq, const void* Wq, void* C, const float* sa, const float* sw, const float* bias,
                                    int epilogue, const float* gate, const void* res, int M, int N, int K, int lda, int ldw, int ldc,
                                    int ldres, int gate_rows, hipStream_t stream) {
  return fp8w4_launch(false, Aq, Wq, C, sa, sw, bias, epilogue, gate, res, M, N, K, lda, ldw, ldc, ldres, gate_rows, stream);
}

/* The MX form (header of this file): Aq [M][lda], Wq [N][ldw] e4m3 bytes; sa8 / sw8 E8M0 bytes in the tiled order
 * [ceil(rows / 128)][K / 128][4][16][8] (byte of row r, elements [128 t + 32 g, + 32): ((r / 128 * K/128 + t) * 4 + g) * 128 + (r % 16) * 8 +
 * (r / 16) % 8 = exponent + 127), as ce_quant_rows_mxfp8 / ce_ln_affine_mxfp8 write them. */
CE_API int ce_gemm_mxfp8(const void* Aq, const void* Wq, void* C, const void* sa8, const void* sw8, const float* bias, int epilogue,
                             const float* gate, const void* res, int M, int N, int K, int lda, int ldw, int ldc, int ldres, int gate_rows,
                             hipStream_t stream) {
  if (!Aq || !Wq || !C || !sa8 || !sw8) return CE_ERR_ARG;
  if (M <= 0 || N <= 0 || K <= 0 || (K % (2 * BKB)) || (N & 7)) return CE_ERR_SHAPE;
  if ((lda & 15) || (ldw & 15) || (ldc & 7)) return CE_ERR_ALIGN;
  if ((long long)M * lda >= (1ll << 32) || (long long)N * ldw >= (1ll << 32)) return CE_ERR_SHAPE;  // 32-bit DMA offsets
  if (epilogue == EPI_GATE_RES && (!res || (ldres & 7))) return CE_ERR_ARG;
  if (epilogue != EPI_BIAS && epilogue != EPI_BIAS_GELU && epilogue != EPI_GATE_RES) return CE_ERR_ARG;
  return fp8w4_launch(true, Aq, Wq, C, reinterpret_cast<const float*>(sa8), reinterpret_cast<const float*>(sw8), bias, epilogue, gate, res, M, N, K,
                      lda, ldw, ldc, ldres, gate_rows, stream);
}

/* bias + tanh GELU with the MX quantisation of the result fused into the epilogue (include/chronoedit_hip.h); the split-K tail of a
 * partially filled last round goes through gemm_fp8w4_reduce_gelu_q. */
CE_API int ce_gemm_mxfp8_gelu_quant(const void* Aq, const void* Wq, const void* sa8, const void* sw8, const float* bias, void* q_out, void* qs_out,
                                        int M, int N, int K, int lda, int ldw, int ldq, hipStream_t stream) {
  if (!Aq || !Wq || !sa8 || !sw8 || !q_out || !qs_out) return CE_ERR_ARG;
  if (M <= 0 || N <= 0 || K <= 0 || (K % (2 * BKB)) || (N & 127)) return CE_ERR_SHAPE;
  if ((lda & 15) || (ldw & 15) || (ldq & 7)) return CE_ERR_ALIGN;
  if ((long long)M * lda >= (1ll << 32) || (long long)N * ldw >= (1ll << 32)) return CE_ERR_SHAPE;
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const int nwg = tiles_m * tiles_n, kt = K / BKB;
  float* g_ws = nullptr;
  size_t g_ws_bytes = 0;
  int g_cus = 256;
  ce_gemm256_workspace(stream, &g_ws, &g_ws_bytes, &g_cus);
  int tail = nwg % g_cus, split = 1;
  if (tail > 0 && g_ws != nullptr) {  // (the heuristics of fp8w4_launch)
    for (int sp = std::min(g_cus / tail, 8); sp >= 2; --sp)
      if (kt % (2 * sp) == 0 && (size_t)tail * sp * BM * BN * sizeof(float) <= g_ws_bytes) {
        split = sp;
        break;
      }
    if (split > 1) {
      const double saving_us = (1.0 - 1.0 / split) * kt * 1.7, slabs_us = (double)tail * split * 0.128 + 8.0;
      if (saving_us < 1.5 * slabs_us) split = 1;
    }
  }
  if (split == 1) tail = 0;
  const int t_full = nwg - tail;
  static bool done_[CE_MAX_DEVICES] = {};
  bool& done = done_[ce_device_slot()];
  if (!done) {
    if (hipFuncSetAttribute((const void*)gemm_fp8_w4<EPI_BIAS_GELU_Q, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) return CE_ERR_ARG;
    if (hipFuncSetAttribute((const void*)gemm_fp8_w4<EPI_BIAS_GELU_Q, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) return CE_ERR_ARG;
    done = true;
  }
  if ((long long)M * ldq < (1ll << 32))
    hipLaunchKernelGGL((gemm_fp8_w4<EPI_BIAS_GELU_Q, true>), dim3(t_full + tail * split), dim3(256), LDS_BYTES, stream, (const unsigned char*)Aq,
                       (const unsigned char*)Wq, (bf16*)q_out, reinterpret_cast<const float*>(sa8), reinterpret_cast<const float*>(sw8), bias, nullptr,
                       nullptr, M, N, K, lda, ldw, ldq, 0, 0, tiles_m, tiles_n, t_full, split, g_ws, (unsigned char*)qs_out);
  else  // (a q_out beyond 32-bit byte offsets: 64-bit store addresses)
    hipLaunchKernelGGL((gemm_fp8_w4<EPI_BIAS_GELU_Q, true, false>), dim3(t_full + tail * split), dim3(256), LDS_BYTES, stream, (const unsigned char*)Aq,
                       (const unsigned char*)Wq, (bf16*)q_out, reinterpret_cast<const float*>(sa8), reinterpret_cast<const float*>(sw8), bias, nullptr,
                       nullptr, M, N, K, lda, ldw, ldq, 0, 0, tiles_m, tiles_n, t_full, split, g_ws, (unsigned char*)qs_out);
  if (tail)
    hipLaunchKernelGGL(gemm_fp8w4_reduce_gelu_q, dim3(4 * tail), dim3(256), 0, stream, (unsigned char*)q_out, (unsigned char*)qs_out, bias, M, N, ldq,
                       tiles_m, tiles_n, t_full, split, g_ws);
  return (int)hipGetLastError();
}

/* How this library was compiled (include/chronoedit_hip.h): bit 0 the diagnostic build (-DCE_DIAGNOSTICS), bit 1 F8_A3, bits 4-7
 * F8_DMA_SCHED, bit 2 F8_EPI_LDS (the staged epilogue of rounds 3-5: W scales in the A order), bits 8-15 F8_ABLATE (non-zero: a timing-only build of the MX fp8 GEMM whose RESULTS ARE GARBAGE - the loader refuses it
 * unless asked for by name). */
CE_API int ce_build_info(void) {
  int v = 0;
#ifdef CE_DIAGNOSTICS
  v |= 1;
#endif
  v |= (F8_A3 ? 2 : 0) | (F8_EPI_LDS ? 4 : 0) | ((F8_DMA_SCHED & 15) << 4) | ((F8_ABLATE & 255) << 8);
  return v;
}
