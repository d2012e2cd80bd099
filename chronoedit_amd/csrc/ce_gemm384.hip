// 384x256x64 bf16 GEMM, one wave per SIMD: 4 waves (2 x 2), wave tile 192 x 128 - the macro tile the register file allows at most.
// Same contract, epilogues, split-K tail and segmented operands as ce_gemm256.hip / ce_gemm256w4.hip (`ce_set_gemm_variant(6)`).
//
// Why: ce_gemm256w4.hip is bounded by energy per flop (76 % MFMA busy at a power-limited 1.78 GHz); what is
// left is the bytes moved per flop.  A 384 x 256 tile needs (384 + 256) operand rows per 384 * 256 outputs: -17 % LDS-DMA / L2 /
// fabric bytes per flop against 256 x 256, 12 + 8 fragment reads per 96 MFMAs and k-step instead of 8 + 8 per 64 (-17 %), and ONE
// workgroup barrier per K-tile of 192 MFMAs per wave.
//  * Accumulators: 12 x 8 tiles of v_mfma_f32_16x16x32_bf16 = 384 registers per lane: row fragments 0..7 in the 256 AGPRs, 8..11
//    in 128 VGPRs (asm MFMAs, "+a" / "+v": D tied to C in place).  The other 128 VGPRs: TWO W fragment sets (k-step 0 / 1: 64), a
//    ring of FOUR A fragments read from LDS three MFMA groups ahead of their use (16), addresses.
//  * LDS: two stages of [A 384 rows | W 256 rows] x 128 B = 160 KiB; 16-byte chunk c of row r in slot c ^ ((r >> 1) & 7) on the
//    source side of the LDS-DMA and on the read side.  Tile t computes from stage t % 2 while tile t + 1 ... t + 2 arrive.
//  * A K-tile = 24 groups of 8 MFMAs (k-step G / 12, row fragment G % 12 against the 8 W fragments of the k-step).  Group G issues,
//    between its MFMAs: the A fragment of group G + 3 (ring slot (G + 3) % 4), one W fragment of the tile's second k-step
//    (groups 1..8) or of the NEXT tile's first k-step (groups 21..23), and LDS-DMA pieces of tile t + 2 (groups 21..23 and, in the
//    next tile, 0..13: 12 A + 8 W pieces of 1 KiB per wave).
//    The ONE barrier of a tile sits between groups 20 and 21, behind `vmcnt(0)` + `lgkmcnt(0)`: every fragment read of tile t has
//    been issued and has completed by then (=> stage t % 2 may be overwritten by tile t + 2 from here on), and tile t + 1 has landed
//    (its last piece was issued seven groups = ~900 MFMA cycles earlier) (=> its fragments may be read from here on).
#include <algorithm>

#include "ce_common.h"
#include "ce_gemm_epi.h"

namespace {

// NF = 16-row fragments of a wave's tile: 12 (384 x 256 macro tile, 192 x 128 wave tiles) or - round 6 - 9 (288 x 256, 144 x 128: M = 7 200, the
// distilled B = 1 step, is 25 tiles of 288 rows exactly where it is 18.75 of 384: at N = 5120 that is 500 workgroups = 1.95 rounds of 256 CUs
// instead of 380 = 1.48 rounds whose tail is cut along K through fp32 slabs; same main loop, 18 groups per K-tile instead of 24)
constexpr int BN = 256, BK = 64;
constexpr int W_TILE = BN * BK * 2;       // 32 KiB
template <int NF> struct T384 {
  static constexpr int BM = 32 * NF;                // 384 | 288
  static constexpr int A_TILE = BM * BK * 2;        // 48 | 36 KiB
  static constexpr int STAGE = A_TILE + W_TILE;     // 80 | 68 KiB
  static constexpr int NG = 2 * NF;                 // groups of 8 MFMAs per K-tile
  static constexpr int NP = NF + 8;                 // LDS-DMA pieces per wave and K-tile (NF of A, 8 of W)
};
constexpr int CROW = BN * 2 + 16;         // padded epilogue staging row (528 B)
constexpr int QROW = 128 * 2 + 16;        // padded staging row of a quadrant (split-K reduce)

typedef __attribute__((address_space(3))) void lds_void;

#define X_BAR() __builtin_amdgcn_s_barrier()
#define X_PIN() __builtin_amdgcn_sched_barrier(0)

template <int BM>
__device__ __forceinline__ void tile_origin_384(int wg, int tiles_m, int tiles_n, int& m0, int& n0) {
  constexpr int GROUP = 4;
  const int group_sz = GROUP * tiles_n;
  const int gid = wg / group_sz;
  const int first_m = gid * GROUP;
  const int gm = min(tiles_m - first_m, GROUP);
  m0 = (first_m + (wg % group_sz) % gm) * BM;
  n0 = ((wg % group_sz) / gm) * BN;
}

// one MFMA: row fragment F (operand a) against column fragment G (operand b); rows 0..7 accumulate in AGPRs, 8..11 in VGPRs.
// Round 6: the A fragment is the FIRST operand - the accumulator holds C (lane (fr, fg): rows 16 F + 4 fg + [0,4), the output column of W
// fragment row fr), not C^T: see the register-direct epilogue.  (Same products, same k order: bit-identical sums.)
template <int F>
__device__ __forceinline__ void mma1(f32x4& acc, const bf16x8& b, const bf16x8& a) {
  if constexpr (F < 8)
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
  else
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

template <int EPI, int NF = 12>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_bf16_384(
    const bf16* __restrict__ A, const bf16* __restrict__ W, bf16* __restrict__ C, const float* __restrict__ bias,
    const float* __restrict__ gate, const bf16* __restrict__ res, int M, int N, int K, int lda, int ldw, int ldc, int ldres,
    int gate_rows, int tiles_m, int tiles_n, int t_full, int split, float* __restrict__ ws, uint32_t a_seg_magic,
    uint32_t a_seg_extra, uint32_t w_seg_magic, uint32_t w_seg_extra) {
  constexpr int BM = T384<NF>::BM, A_TILE = T384<NF>::A_TILE, STAGE = T384<NF>::STAGE, NG = T384<NF>::NG, NP = T384<NF>::NP;
  static_assert(NF >= 9 && NF <= 12, "the W fragments of the second k-step are read in groups 1..8 of the first; rows 8.. of the accumulator live in 128 VGPRs");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fg = lane >> 4;

#ifdef G384_STAGGER_CYC  // diagnostic builds only: the first round's workgroups start in four phases G384_STAGGER_CYC shader cycles apart, so
  // that the rounds of a launch stop finishing (and storing their C tiles) on all CUs at the same moment
  if (blockIdx.x < 256) {
    const long long t0 = __builtin_readcyclecounter();
    const long long wait = (long long)((blockIdx.x >> 3) & 3) * G384_STAGGER_CYC;
    while (__builtin_readcyclecounter() - t0 < wait) __builtin_amdgcn_s_sleep(8);
  }
#endif
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
  tile_origin_384<BM>(wg, tiles_m, tiles_n, m0, n0);
  const int kt_last = ktn - 1;

  // LDS-DMA sources.  Piece p of this wave = rows 8 (wave + 4 p) .. + 8 of the operand tile (lane l: row + (l >> 3), slot l & 7 <- chunk
  // slot ^ ((row >> 1) & 7), the same for every p).  The per-lane byte offset of piece p is min(off0 + p * 32 rows, last row): two
  // VALU ops per piece instead of 20 resident offset registers (rows past the operand's end re-read its last row; never stored).
  const int prow = 8 * wave + (lane >> 3);
  const uint32_t pchunk = (uint32_t)(((lane & 7) ^ ((prow >> 1) & 7)) << 4);
  const uint32_t a_off0 = (uint32_t)(m0 + prow) * (uint32_t)(lda * 2) + pchunk, a_lim = (uint32_t)(M - 1) * (uint32_t)(lda * 2) + pchunk;
  // W rows are PERMUTED on the source side (round 6): LDS row 16 G + i of a wave's 128 W rows (fragment G, fragment row i) holds W row 8 i + G,
  // so that a lane's eight G accumulators of one output row are eight CONSECUTIVE columns.  LDS row r = 32 j + prow of the tile (piece j) <-
  // W row (r & 128) + 8 (r & 15) + ((r & 127) >> 4) = 128 (j >> 2) + 2 (j & 3) + [8 (prow & 15) + (prow >> 4)]: a per-lane base + a per-piece constant
  const uint32_t w_rowb = (uint32_t)(ldw * 2);
  const uint32_t w_off0 = (uint32_t)(n0 + ((prow & 15) << 3) + (prow >> 4)) * w_rowb + pchunk, w_lim = (uint32_t)(N - 1) * w_rowb + pchunk;
  const uint32_t a_pstride = (uint32_t)(32 * lda * 2);
  const auto a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, 0xffffffffu, 0x00020000);
  const auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, 0xffffffffu, 0x00020000);
  auto koff = [&](int t, uint32_t magic, uint32_t extra) __attribute__((always_inline)) -> int {
    const int ta = kt0 + min(t, kt_last);
    return ta * (BK * 2) + (int)((((uint32_t)ta * magic) >> 16) * extra);
  };
  // piece q of a tile's NP per wave: 0..NF-1 = A pieces, NF..NP-1 = W pieces
  auto dma = [&](int q, int stage_bytes, int a_soff, int w_soff) __attribute__((always_inline)) {
    if (q < NF) {
      const uint32_t v = min(a_off0 + (uint32_t)q * a_pstride, a_lim);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_void*)(smem + stage_bytes + (wave + 4 * q) * 1024), 16, v, a_soff, 0, 0);
    } else {
      const uint32_t v = min(w_off0 + (uint32_t)(128 * ((q - NF) >> 2) + 2 * ((q - NF) & 3)) * w_rowb, w_lim);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void*)(smem + stage_bytes + A_TILE + (wave + 4 * (q - NF)) * 1024), 16, v, w_soff, 0, 0);
    }
  };

  // fragment read addresses [stage parity][k-step]: row (wm*192 | wn*128) + f*16 + fr, chunk (fg + 4 ks) ^ (fr >> 1);  + f*2048
  int a_rd[2][2], w_rd[2][2];
#pragma unroll
  for (int par = 0; par < 2; ++par)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      a_rd[par][ks] = par * STAGE + (wm * (16 * NF) + fr) * 128 + (((fg + 4 * ks) ^ (fr >> 1)) << 4);
      w_rd[par][ks] = par * STAGE + A_TILE + (wn * 128 + fr) * 128 + (((fg + 4 * ks) ^ (fr >> 1)) << 4);
    }

  f32x4 acc[NF][8];
#pragma unroll
  for (int f = 0; f < NF; ++f)
#pragma unroll
    for (int g = 0; g < 8; ++g) acc[f][g] = f32x4{0.f, 0.f, 0.f, 0.f};

  // prologue: tile 0 and the first three pieces of tile 1 on their way; tile 0 landed; W fragments of its first k-step and the A fragments of its groups 0..2 read
  {
    const int a0 = koff(0, a_seg_magic, a_seg_extra), w0 = koff(0, w_seg_magic, w_seg_extra);
    const int a1 = koff(1, a_seg_magic, a_seg_extra), w1 = koff(1, w_seg_magic, w_seg_extra);
#pragma unroll
    for (int q = 0; q < NP; ++q) dma(q, 0, a0, w0);
#pragma unroll
    for (int q = 0; q < 3; ++q) dma(q, STAGE, a1, w1);  // (pieces 3..19 of tile 1 follow in groups 0..13 of tile 0, as in steady state)
  }
#ifndef G384_ABLATE_PROLOGUE  // (diagnostic builds only, tools/gemm_ab.py: what a tile costs without waiting for its first K-tile - garbage results -
                             //  = what overlapping the next tile's prologue with this tile's epilogue could buy at most)
  asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
#endif
  X_BAR();
  bf16x8 ring[4], bx[8], by[8];
#pragma unroll
  for (int g = 0; g < 8; ++g) bx[g] = *reinterpret_cast<const bf16x8*>(smem + w_rd[0][0] + g * 2048);
#pragma unroll
  for (int f = 0; f < 3; ++f) ring[f] = *reinterpret_cast<const bf16x8*>(smem + a_rd[0][0] + f * 2048);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  X_PIN();

  // ring slot of group g of a tile in stage par: the ring runs on across tiles, so the slot follows the group count of the PAIR of tiles
  // (NG = 24: the same slots in both stages; NG = 18: the odd stage is two slots on)
#define RS_(g, par) (((g) + (par) * NG) % 4)
  // group G (0..NG-1) of the tile in stage PAR; T = index of that tile (runtime); fillers between the MFMAs, one scheduling region each
#define X_GROUP(G, PAR, T)                                                                                                       \
  {                                                                                                                              \
    constexpr int ks_ = (G) / NF, f_ = (G) % NF;                                                                                 \
    constexpr int gn_ = (G) + 3;                          /* the group whose A fragment is fetched now */                        \
    constexpr bool nxt_ = gn_ >= NG;                      /* ... in the next tile (other stage; legal: G >= NG - 3 is behind the barrier) */ \
    constexpr int ksn_ = (nxt_ ? gn_ - NG : gn_) / NF, fn_ = (nxt_ ? gn_ - NG : gn_) % NF;                                       \
    if ((G) == NG - 3) {                                                                                                         \
      asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");                                                   \
      X_BAR();                                                                                                                   \
      X_PIN();                                                                                                                   \
    }                                                                                                                            \
    bf16x8(&bb_)[8] = ks_ ? by : bx;                                                                                             \
    mma1<f_>(acc[f_][0], bb_[0], ring[RS_(G, PAR)]);                                                                                 \
    ring[RS_(gn_, PAR)] = *reinterpret_cast<const bf16x8*>(smem + a_rd[nxt_ ? 1 - (PAR) : (PAR)][ksn_] + fn_ * 2048);            \
    X_PIN();                                                                                                                     \
    mma1<f_>(acc[f_][1], bb_[1], ring[RS_(G, PAR)]);                                                                                 \
    if ((G) >= 1 && (G) <= 8) by[((G) >= 1 && (G) <= 8) ? (G) - 1 : 0] = *reinterpret_cast<const bf16x8*>(smem + w_rd[(PAR)][1] + ((G) - 1) * 2048);          \
    if ((G) == NG - 3) { bx[0] = *reinterpret_cast<const bf16x8*>(smem + w_rd[1 - (PAR)][0] + 0 * 2048);                          \
                     bx[1] = *reinterpret_cast<const bf16x8*>(smem + w_rd[1 - (PAR)][0] + 1 * 2048);                              \
                     bx[2] = *reinterpret_cast<const bf16x8*>(smem + w_rd[1 - (PAR)][0] + 2 * 2048); }                            \
    if ((G) == NG - 2) { bx[3] = *reinterpret_cast<const bf16x8*>(smem + w_rd[1 - (PAR)][0] + 3 * 2048);                          \
                     bx[4] = *reinterpret_cast<const bf16x8*>(smem + w_rd[1 - (PAR)][0] + 4 * 2048);                              \
                     bx[5] = *reinterpret_cast<const bf16x8*>(smem + w_rd[1 - (PAR)][0] + 5 * 2048); }                            \
    X_PIN();                                                                                                                     \
    mma1<f_>(acc[f_][2], bb_[2], ring[RS_(G, PAR)]);                                                                                 \
    if ((G) == NG - 1) { bx[6] = *reinterpret_cast<const bf16x8*>(smem + w_rd[1 - (PAR)][0] + 6 * 2048);                          \
                     bx[7] = *reinterpret_cast<const bf16x8*>(smem + w_rd[1 - (PAR)][0] + 7 * 2048); }                            \
    X_PIN();                                                                                                                     \
    mma1<f_>(acc[f_][3], bb_[3], ring[RS_(G, PAR)]);                                                                                 \
    /* LDS-DMA of the tile two ahead of the one whose stage is being overwritten: pieces 0..2 in groups 21..23 (into THIS tile's   \
       stage, free behind the barrier), pieces 3..19 in groups 0..13 of the next tile (= into the other stage, seen from there) */ \
    if ((G) >= NG - 3) dma((G) - (NG - 3), (PAR) * STAGE, a_soff_next, w_soff_next);                                             \
    if ((G) <= 2) { dma(3 + 2 * (G), (1 - (PAR)) * STAGE, a_soff_prev, w_soff_prev); }                                            \
    if ((G) >= 3 && (G) <= NP - 7) dma(6 + (G), (1 - (PAR)) * STAGE, a_soff_prev, w_soff_prev);                                    \
    X_PIN();                                                                                                                     \
    mma1<f_>(acc[f_][4], bb_[4], ring[RS_(G, PAR)]);                                                                                 \
    if ((G) <= 2) { dma(4 + 2 * (G), (1 - (PAR)) * STAGE, a_soff_prev, w_soff_prev); }                                            \
    X_PIN();                                                                                                                     \
    mma1<f_>(acc[f_][5], bb_[5], ring[RS_(G, PAR)]);                                                                                 \
    mma1<f_>(acc[f_][6], bb_[6], ring[RS_(G, PAR)]);                                                                                 \
    mma1<f_>(acc[f_][7], bb_[7], ring[RS_(G, PAR)]);                                                                                 \
    X_PIN();                                                                                                                     \
  }
  // pieces of groups 0..13: 3,4 | 5,6 | 7,8 | 9 .. 19  (two per group in groups 0..2, one per group in 3..13: 6 + 11 = 17)
#define X_TILE(PAR, T)                                                                                                           \
  {                                                                                                                              \
    /* groups 0..13 finish the transfer of tile T + 1 (started in groups 21..23 of tile T - 1, into stage 1 - PAR);              \
       groups 21..23 start the transfer of tile T + 2 (into stage PAR) */                                                       \
    const int a_soff_prev = koff((T) + 1, a_seg_magic, a_seg_extra), w_soff_prev = koff((T) + 1, w_seg_magic, w_seg_extra);       \
    const int a_soff_next = koff((T) + 2, a_seg_magic, a_seg_extra), w_soff_next = koff((T) + 2, w_seg_magic, w_seg_extra);       \
    X_GROUP(0, PAR, T) X_GROUP(1, PAR, T) X_GROUP(2, PAR, T) X_GROUP(3, PAR, T) X_GROUP(4, PAR, T) X_GROUP(5, PAR, T)             \
    X_GROUP(6, PAR, T) X_GROUP(7, PAR, T) X_GROUP(8, PAR, T) X_GROUP(9, PAR, T) X_GROUP(10, PAR, T) X_GROUP(11, PAR, T)           \
    X_GROUP(12, PAR, T) X_GROUP(13, PAR, T) X_GROUP(14, PAR, T) X_GROUP(15, PAR, T) X_GROUP(16, PAR, T) X_GROUP(17, PAR, T)       \
    if constexpr (NG > 18) {                                                                                                     \
      X_GROUP(18, PAR, T) X_GROUP(19, PAR, T) X_GROUP(20, PAR, T) X_GROUP(21, PAR, T) X_GROUP(22, PAR, T) X_GROUP(23, PAR, T)     \
    }                                                                                                                            \
  }

  const int npairs = ktn >> 1;
  for (int it = 0; it < npairs; ++it) {
    const int t = 2 * it;
    X_TILE(0, t)
    X_TILE(1, t + 1)
  }
#undef X_TILE
#undef X_GROUP
#undef RS_
  asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");

  // ---- epilogue, REGISTER-DIRECT (round 6; the fp8 GEMM's, csrc/ce_gemm_fp8w4.hip).  The accumulator holds C: lane (fr, fg) of acc[f][g] owns rows
  // f*16 + fg*4 + [0,4) of the wave tile and - by the DMA's source-side row permutation - output column 8 fr + g.  For one row (f, jj) a lane's eight
  // g accumulators are EIGHT CONSECUTIVE COLUMNS: one 16-byte store per lane, the sixteen fr lanes of a quad-row cover 256 contiguous bytes of the
  // row (two whole cache lines), four rows per wave instruction - the row-contiguous stores the LDS staging of rounds 3-5 existed for, without the
  // LDS round trip and its twelve barriers.  (Round 4's direct form kept C^T and stored 16 rows x 64 bytes per instruction: level.)  Same
  // arithmetic and roundings as the staged form: bit-identical results.
  const int col0 = n0 + wn * 128 + fr * 8;  // this lane's 8 output columns
  const bool col_ok = col0 < N;
  const int colc = min(col0, N - 8);
  const int row_base = m0 + wm * (16 * NF) + fg * 4;  // + f*16 + jj
  if (partial) {
    // split-K tail piece: fp32 slab in the [wave][f][g][lane'] order gemm384_reduce reads (the C^T accumulator order of the staged form: lane'
    // (fr', fg') of [f][g] = row f*16 + fr', columns g*16 + fg'*4 + [0,4)): this lane's (f, jj, g = 4 h .. 4 h + 3) is row f*16 + fg*4 + jj,
    // columns fr*8 + 4 h + [0,4) -> fr' = fg*4 + jj, g' = fr >> 1, fg' = 2 (fr & 1) + h
    float* slab = ws + (size_t)(blockIdx.x - t_full) * (BM * BN);
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      f32x4 av[8];
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        if (f < 8) asm volatile("" : "+a"(acc[f][g])); else asm volatile("" : "+v"(acc[f][g]));  // (pins the read-out of fragment row f here)
        av[g] = acc[f][g];
      }
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x4 o = {av[4 * h + 0][jj], av[4 * h + 1][jj], av[4 * h + 2][jj], av[4 * h + 3][jj]};
          const int lane_o = fg * 4 + jj + 16 * (2 * (fr & 1) + h);
          *reinterpret_cast<f32x4*>(slab + (((wave * (8 * NF) + f * 8 + (fr >> 1)) * 64) + lane_o) * 4) = o;
        }
    }
    return;
  }
#ifdef G384_ABLATE_EPILOGUE  // diagnostic builds only (tools/gemm_ab.py): what the tile costs without its epilogue (results are not written)
  if (M > 0) return;
#endif
  f32x4 bvv[2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
    bvv[h] = (EPI != EPI_BIAS_ROW && bias != nullptr) ? *reinterpret_cast<const f32x4*>(bias + colc + 4 * h) : f32x4{0.f, 0.f, 0.f, 0.f};
  if constexpr (EPI == EPI_BIAS_T) {
    // The TRANSPOSE of the tile is stored (C is [N][ldc]): for output column col0 + g a lane owns rows row_base + f*16 + [0,4) - four consecutive
    // elements of row col0 + g of C^T.  (M % 4 == 0: the launcher checks; never a split-K tail piece.)
    const auto t_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)C, 0, (uint32_t)(N - 1) * (uint32_t)(ldc * 2) + (uint32_t)M * 2u, 0x00020000);
    if constexpr (NF == 12) {
      // Stored straight from the registers (8 bytes per lane: sixteen C^T rows x 32 bytes per instruction) the launch wrote 233 MB for a 147 MB
      // result - quarter lines leave the L2 before their neighbours arrive (PMC WRITE_SIZE, profiles/r06_pmc_vt_store.txt) - and the better tile
      // fit bought 4 % where 10 % were on the table.  So each wave turns its 128 x 192 sub-tile through a private LDS region in three passes of
      // four row fragments: 64 consecutive rows of the tile = exactly one aligned 128-byte line of every C^T row (a wave's first row is a
      // multiple of 192), and a store instruction writes eight WHOLE lines.  One workgroup barrier first: slower waves still read the stages.
      constexpr int TROW = 128 + 16;  // padded LDS row: 64 elements
      X_BAR();
      unsigned char* const tb = smem + wave * (128 * TROW);
      const int n_base = n0 + wn * 128, m_base = m0 + wm * 192;
#pragma unroll
      for (int pass = 0; pass < 3; ++pass) {
#pragma unroll
        for (int ff = 0; ff < 4; ++ff) {
          const int f = 4 * pass + ff;
          f32x4 av[8];
#pragma unroll
          for (int g = 0; g < 8; ++g) {
            if (f < 8) asm volatile("" : "+a"(acc[f][g])); else asm volatile("" : "+v"(acc[f][g]));  // (pins the read-out of fragment row f here)
            av[g] = acc[f][g];
          }
#pragma unroll
          for (int g = 0; g < 8; ++g) {
            const float b = bvv[g >> 2][g & 3];
            const u32x2 y = {pack_bf16(av[g][0] + b, av[g][1] + b), pack_bf16(av[g][2] + b, av[g][3] + b)};
            *reinterpret_cast<u32x2*>(tb + (fr * 8 + g) * TROW + (ff * 16 + fg * 4) * 2) = y;
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (wave-private region: no barrier)
#pragma unroll
        for (int i = 0; i < 16; ++i) {  // 128 rows x 8 chunks of 16 bytes: lane -> (row 8 i + (lane >> 3), chunk lane & 7)
          const int r = 8 * i + (lane >> 3), c = lane & 7;
          const u32x4 v = *reinterpret_cast<const u32x4*>(tb + r * TROW + c * 16);
          const int n = n_base + r, m = m_base + pass * 64 + c * 8;
          const uint32_t toff = (n < N && m < M) ? (uint32_t)n * (uint32_t)(ldc * 2) + (uint32_t)m * 2u : 0xffffffffu;
          __builtin_amdgcn_raw_buffer_store_b128(v, t_rsrc, toff, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the reads are done before the next pass overwrites the region)
      }
      return;
    }
    // (NF = 9: a wave's 144 rows are 288 bytes of a C^T row - no whole number of lines whatever the split; the direct form)
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      f32x4 av[8];
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        if (f < 8) asm volatile("" : "+a"(acc[f][g])); else asm volatile("" : "+v"(acc[f][g]));  // (pins the read-out of fragment row f here)
        av[g] = acc[f][g];
      }
      const int m = row_base + f * 16;
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const float b = bvv[g >> 2][g & 3];
        const u32x2 y = {pack_bf16(av[g][0] + b, av[g][1] + b), pack_bf16(av[g][2] + b, av[g][3] + b)};
        const uint32_t toff = (m < M && col_ok) ? (uint32_t)(col0 + g) * (uint32_t)(ldc * 2) + (uint32_t)m * 2u : 0xffffffffu;
        __builtin_amdgcn_raw_buffer_store_b64(y, t_rsrc, toff, 0, 0);
      }
    }
    return;
  }
  // (the launcher sends C beyond 32-bit byte offsets, and gate rows shorter than a tile, to the 8-wave kernel)
  const auto c_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)C, 0, (uint32_t)(M - 1) * (uint32_t)(ldc * 2) + (uint32_t)N * 2u, 0x00020000);
  f32x4 gA[2], gB[2];
  int g_switch = 0x7fffffff;
  constexpr int RA = 2;  // row fragments of residual in flight ahead of the one being finished (4 x 16 B per lane each)
  u32x4 rv[RA + 1][4];
  auto res_load = [&](int f, u32x4* dst) __attribute__((always_inline)) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) dst[jj] = *reinterpret_cast<const u32x4*>(res + (size_t)min(row_base + f * 16 + jj, M - 1) * ldres + colc);
  };
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
    for (int f = 0; f < RA; ++f) res_load(f, rv[f % (RA + 1)]);
  }
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    if (EPI == EPI_GATE_RES && f + RA < NF) res_load(f + RA, rv[(f + RA) % (RA + 1)]);
    f32x4 av[8];  // the eight accumulators of this fragment row, whole
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      if (f < 8) asm volatile("" : "+a"(acc[f][g])); else asm volatile("" : "+v"(acc[f][g]));  // (pins the read-out of fragment row f here)
      av[g] = acc[f][g];
    }
    f32x4 brow4 = {0.f, 0.f, 0.f, 0.f};
    if (EPI == EPI_BIAS_ROW) brow4 = *reinterpret_cast<const f32x4*>(bias + min(row_base + f * 16, M - 4));  // (M % 4 == 0: the launcher checks)
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
        const u32x4 r = rv[f % (RA + 1)][jj];
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
}

// Sums the `split` fp32 slabs of one wave quadrant (192 x 128 accumulators) of a tail tile and applies the epilogue.
// grid = 4 x the number of tail tiles, 256 threads: thread (w, lane) takes accumulator rows f = 3w .. 3w + 2 of the quadrant.
template <int EPI, int NF = 12>
__global__ __launch_bounds__(256) void gemm384_reduce(bf16* __restrict__ C, const float* __restrict__ bias, const float* __restrict__ gate,
                                                      const bf16* __restrict__ res, int M, int N, int ldc, int ldres, int gate_rows,
                                                      int tiles_m, int tiles_n, int t_full, int split, const float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int tile = blockIdx.x >> 2, q = blockIdx.x & 3;  // q = producer wave = (wm, wn)
  constexpr int BM = T384<NF>::BM;
  int m0, n0;
  tile_origin_384<BM>(t_full + tile, tiles_m, tiles_n, m0, n0);
  m0 += (q >> 1) * (16 * NF);
  n0 += (q & 1) * 128;
  const float* slab = ws + (size_t)tile * split * (BM * BN);
#pragma unroll
  for (int ff = 0; ff < 3; ++ff) {
    const int f = 3 * w + ff;  // (NF = 9: the fourth wave has no rows)
    if (f >= NF) break;
    const int rl = f * 16 + fr;
    float brow = 0.f;
    if (EPI == EPI_BIAS_ROW) brow = bias[min(m0 + rl, M - 1)];
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const int cl = g * 16 + fg * 4;
      f32x4 bv = (EPI != EPI_BIAS_ROW && bias != nullptr) ? *reinterpret_cast<const f32x4*>(bias + min(n0 + cl, N - 4)) : f32x4{0.f, 0.f, 0.f, 0.f};
      if (EPI == EPI_BIAS_ROW) bv[0] = bv[1] = bv[2] = bv[3] = brow;
      const int e = (((q * (8 * NF) + f * 8 + g) * 64) + lane) * 4;
      f32x4 v = *reinterpret_cast<const f32x4*>(slab + e);
      for (int sidx = 1; sidx < split; ++sidx) v += *reinterpret_cast<const f32x4*>(slab + (size_t)sidx * (BM * BN) + e);
      const u32x2 pk = {pack_bf16(v[0] + bv[0], v[1] + bv[1]), pack_bf16(v[2] + bv[2], v[3] + bv[3])};
      *reinterpret_cast<u32x2*>(smem + rl * QROW + cl * 2) = pk;
    }
  }
  __syncthreads();
  epi_chunks<EPI, NF>(smem, QROW, [&](int tt, int& rl, int& cc, int& mr) { const int c = tid + 256 * tt; rl = mr = c >> 4; cc = c & 15; }, m0, n0,
                      C, gate, res, M, N, ldc, ldres, gate_rows);
}

}  // namespace

extern "C" void ce_gemm256_workspace(hipStream_t stream, float** ws, size_t* bytes, int* cus);
extern "C" int ce_gemm256_launch(const void* A, const void* W, void* C, const float* bias, int epilogue, const float* gate,
                                 const void* res, int M, int N, int K, int lda, int ldw, int ldc, int ldres, int gate_rows,
                                 int a_seg_k, long long a_seg_stride, int w_seg_k, long long w_seg_stride, hipStream_t stream);

template <int NF>
static int gemm384_launch_impl(const void* A, const void* W, void* C, const float* bias, int epilogue, const float* gate, const void* res, int M, int N,
                               int K, int lda, int ldw, int ldc, int ldres, int gate_rows, int a_seg_k, long long a_seg_stride, int w_seg_k,
                               long long w_seg_stride, hipStream_t stream) {
  constexpr int BM = T384<NF>::BM, STAGE = T384<NF>::STAGE;
  if (epilogue == EPI_BIAS_T) {  // (C is [N][ldc]; no other kernel stores the transpose: the dispatcher only comes here with a shape this one takes)
    if ((M & 7) || (long long)N * ldc * 2 >= (1ll << 32) || !bias) return CE_ERR_SHAPE;
  } else
  if ((epilogue == EPI_GATE_RES && gate != nullptr && gate_rows > 0 && gate_rows < BM) || (long long)M * ldc * 2 >= (1ll << 32) ||
      (epilogue == EPI_BIAS_ROW && (M & 3)))  // (the register-direct epilogue stores through 32-bit buffer offsets and reads four row biases at once)
    return ce_gemm256_launch(A, W, C, bias, epilogue, gate, res, M, N, K, lda, ldw, ldc, ldres, gate_rows, a_seg_k, a_seg_stride, w_seg_k,
                             w_seg_stride, stream);
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
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
    if (extra < 0 || extra * (K / seg_k) + (long long)K * 2 >= (1ll << 31)) return CE_ERR_SHAPE;
    extra_out = (uint32_t)extra;
    return CE_OK;
  };
  if (int rc = seg(a_seg_k, a_seg_stride, a_seg_magic, a_seg_extra)) return rc;
  if (int rc = seg(w_seg_k, w_seg_stride, w_seg_magic, w_seg_extra)) return rc;
  float* g_ws = nullptr;
  size_t g_ws_bytes = 0;
  int g_cus = 256;
  ce_gemm256_workspace(stream, &g_ws, &g_ws_bytes, &g_cus);
  int tail = nwg % g_cus, split = 1;
  if (tail > 0 && g_ws != nullptr && epilogue != EPI_BIAS_T) {  // (the transposed store has no reduce form: its last round runs whole)
    for (int s = std::min(g_cus / tail, 8); s >= 2; --s)
      if (kt % (2 * s) == 0 && (size_t)tail * s * BM * BN * sizeof(float) <= g_ws_bytes) {
        split = s;
        break;
      }
  }
  if (split == 1) tail = 0;
  const int t_full2 = nwg - tail;
  dim3 grid(t_full2 + tail * split), block(256);
  const int lds = 2 * STAGE;
  static bool attr_done_[CE_MAX_DEVICES][8] = {};
  bool* attr_done = attr_done_[ce_device_slot()];
#define CE_LAUNCH(E)                                                                                                       \
  do {                                                                                                                     \
    if (!attr_done[E]) {                                                                                                   \
      if (hipFuncSetAttribute((const void*)gemm_bf16_384<E, NF>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return CE_ERR_ARG; \
      (void)hipFuncSetAttribute((const void*)gemm384_reduce<E, NF>, hipFuncAttributeMaxDynamicSharedMemorySize, 16 * NF * QROW); \
      attr_done[E] = true;                                                                                                 \
    }                                                                                                                      \
    hipLaunchKernelGGL((gemm_bf16_384<E, NF>), grid, block, lds, stream, (const bf16*)A, (const bf16*)W, (bf16*)C, bias, gate, \
                       (const bf16*)res, M, N, K, lda, ldw, ldc, ldres, gate_rows, tiles_m, tiles_n, t_full2, split, g_ws, \
                       a_seg_magic, a_seg_extra, w_seg_magic, w_seg_extra);                                                \
    if (tail)                                                                                                              \
      hipLaunchKernelGGL((gemm384_reduce<E, NF>), dim3(4 * tail), block, 16 * NF * QROW, stream, (bf16*)C, bias, gate,     \
                         (const bf16*)res, M, N, ldc, ldres, gate_rows, tiles_m, tiles_n, t_full2, split, g_ws);           \
  } while (0)
  switch (epilogue) {
    case EPI_BIAS: CE_LAUNCH(EPI_BIAS); break;
    case EPI_BIAS_GELU: CE_LAUNCH(EPI_BIAS_GELU); break;
    case EPI_GATE_RES: CE_LAUNCH(EPI_GATE_RES); break;
    case EPI_BIAS_GELU_ERF: CE_LAUNCH(EPI_BIAS_GELU_ERF); break;
    case EPI_BIAS_ROW: CE_LAUNCH(EPI_BIAS_ROW); break;
    case EPI_BIAS_T: {
      static bool t_done_[CE_MAX_DEVICES] = {};
      bool& t_done = t_done_[ce_device_slot()];
      if (!t_done) {
        if (hipFuncSetAttribute((const void*)gemm_bf16_384<EPI_BIAS_T, NF>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return CE_ERR_ARG;
        t_done = true;
      }
      hipLaunchKernelGGL((gemm_bf16_384<EPI_BIAS_T, NF>), grid, block, lds, stream, (const bf16*)A, (const bf16*)W, (bf16*)C, bias, gate, (const bf16*)res,
                         M, N, K, lda, ldw, ldc, ldres, gate_rows, tiles_m, tiles_n, t_full2, split, g_ws, a_seg_magic, a_seg_extra, w_seg_magic,
                         w_seg_extra);
      break;
    }
    default: return CE_ERR_ARG;
  }
#undef CE_LAUNCH
  return (int)hipGetLastError();
}

extern "C" int ce_gemm384_launch(const void* A, const void* W, void* C, const float* bias, int epilogue, const float* gate,
                                 const void* res, int M, int N, int K, int lda, int ldw, int ldc, int ldres, int gate_rows,
                                 int a_seg_k, long long a_seg_stride, int w_seg_k, long long w_seg_stride, hipStream_t stream) {
  return gemm384_launch_impl<12>(A, W, C, bias, epilogue, gate, res, M, N, K, lda, ldw, ldc, ldres, gate_rows, a_seg_k, a_seg_stride, w_seg_k,
                                 w_seg_stride, stream);
}

// the 288 x 256 macro tile (NF = 9): same kernel, 144 x 128 wave tiles
extern "C" int ce_gemm288_launch(const void* A, const void* W, void* C, const float* bias, int epilogue, const float* gate,
                                 const void* res, int M, int N, int K, int lda, int ldw, int ldc, int ldres, int gate_rows,
                                 int a_seg_k, long long a_seg_stride, int w_seg_k, long long w_seg_stride, hipStream_t stream) {
  return gemm384_launch_impl<9>(A, W, C, bias, epilogue, gate, res, M, N, K, lda, ldw, ldc, ldres, gate_rows, a_seg_k, a_seg_stride, w_seg_k,
                                w_seg_stride, stream);
}
