// 256x256x64 bf16 GEMM for the large DiT projections — the MI355X-tuned variant of ce_gemm.hip
// (same contract, same epilogues; selected by ce_gemm_bf16 when the shape allows it).
//
// Structure (CDNA4):
//  * 512-thread workgroup, 8 waves as 2(M) x 4(N); wave (wm, wn) owns rows {i*128 + wm*64 + [0,64)}
//    and cols {j*128 + wn*32 + [0,32)}, i,j in {0,1}: four 64x32 "quadrants" Q(i,j), each
//    4x2 v_mfma_f32_16x16x32_bf16 accumulators (128 accumulator VGPRs per lane).
//  * Operand K-tiles (64 deep) live in LDS as eight 16-KiB half-tile slots
//    {even,odd K-tile} x {A rows 0-127, A rows 128-255, W rows 0-127, W rows 128-255}  (128 KiB);
//    with the interleaved row ownership above, quadrant operands map 1:1 onto half-tile slots.
//  * Global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction, no VGPR
//    round trip).  The DMA image is lane-linear, so the 16-B-chunk XOR swizzle that makes the
//    ds_read_b128 fragment reads conflict-free is applied to the per-lane SOURCE address and to
//    the read address (cdna guide rule 21).
//  * 4 phases per K-tile (8 per even/odd pair).  Phase p: [counted vmcnt at phase 4], ONE s_barrier,
//    stage one half-tile of a FUTURE K-tile (2 LDS-DMA per lane), 16 MFMAs under s_setprio 1 with
//    the NEXT phase's fragment reads (ds_read_b128) issued between the two 8-MFMA k-steps, into the
//    operand registers the k-step just released (software pipelining without extra VGPRs):
//        phase of tile T      1          2              3            4
//        MFMA                Q00        Q01            Q11          Q10
//        fragment reads       -     A1(T) -> ra         -     B1,A0,B0 of tile T+1
//        stage           other.A1(T+1)  cur.A0(T+2)  cur.B0(T+2)  cur.B1(T+2)
//        wait                 -          -              -         vmcnt(4)
//    A slot is restaged at the earliest one barrier after the lgkmcnt(0) that completed its last
//    read, and read at the earliest one barrier after the counted vmcnt that retires it; vmcnt(4)
//    leaves the two most recent half-tiles in flight, LDS-DMA is never drained inside the loop.
//  * Epilogue: the accumulators go through LDS in two 128-row passes so that bias/GELU/gated-residual
//    math and the global stores run on 16-B row-contiguous chunks (same rounding points as ce_gemm.hip).
#include <algorithm>

#include <mutex>

#include "ce_common.h"
#include "ce_gemm_epi.h"

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int SLOT = 128 * BK * 2;        // 16 KiB half-tile
constexpr int LDS_TILES = 8 * SLOT;       // 128 KiB
constexpr int CROW = BN * 2 + 16;         // padded epilogue staging row (528 B)
constexpr int LDS_BYTES = LDS_TILES > 128 * CROW ? LDS_TILES : 128 * CROW;

// slot ids
constexpr int S_A0 = 0, S_A1 = 1, S_B0 = 2, S_B1 = 3;  // + 4 for the odd K-tile

__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

struct Stager {
  // per-lane byte offsets (row * ld * 2 + swizzled chunk * 16) of the two DMA rounds of each half-tile
  uint32_t a_off[2][2];  // [half][round]
  uint32_t w_off[2][2];
  const char* a_base;
  const char* w_base;
  int wave;              // wave-uniform
  int kt_last;
  // Segmented A (Ulysses receive layout, ce_gemm_aseg_bf16): K-tile t of A starts a_seg_extra bytes further for every
  // a_seg_tiles tiles passed, i.e. at t*128 + (t / a_seg_tiles) * a_seg_extra; t / a_seg_tiles = (t * a_seg_magic) >> 16
  // (checked on the host for every tile index of the launch).  a_seg_extra == 0: plain row-major A.
  int kt0;               // first K-tile of this block (split-K tail pieces start past 0)
  uint32_t a_seg_magic;
  uint32_t a_seg_extra;  // bytes
  uint32_t w_seg_magic;  // the same for W (ce_gemm_seg_bf16: weights re-packed K-slab-major)
  uint32_t w_seg_extra;
};

template <int SLOT_ID>
__device__ __forceinline__ void stage_half(unsigned char* smem, const Stager& s, int tile) {
  constexpr int half = SLOT_ID & 1;
  constexpr bool isB = (SLOT_ID & 2) != 0;
  const int t = tile < s.kt_last ? tile : s.kt_last;  // clamp: surplus prefetches re-read the last K-tile
  const int ta = s.kt0 + t;  // absolute K-tile
  const char* base = (isB ? s.w_base + (size_t)(((uint32_t)(ta * s.w_seg_magic) >> 16) * s.w_seg_extra)
                          : s.a_base + (size_t)(((uint32_t)(ta * s.a_seg_magic) >> 16) * s.a_seg_extra)) + (size_t)ta * (BK * 2);
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const uint32_t off = isB ? s.w_off[half][r] : s.a_off[half][r];
    unsigned char* dst = smem + SLOT_ID * SLOT + (r * 8 + s.wave) * 1024;  // wave-uniform; lane l lands at +16 l
    __builtin_amdgcn_global_load_lds((gbl_void*)(base + off), (lds_void*)dst, 16, 0, 0);
  }
}

// fragment reads of one k-step (32 deep): 64 rows of an A half-tile -> 4 fragments, 32 rows of a W half-tile -> 2
template <int SLOT_ID, int KS>
__device__ __forceinline__ void read_a(const unsigned char* smem, int wm, int fr, int fg, bf16x8 (&a)[4]) {
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const int row = wm * 64 + f * 16 + fr;
    a[f] = *reinterpret_cast<const bf16x8*>(smem + SLOT_ID * SLOT + row * (BK * 2) + (((fg + 4 * KS) ^ swz(row)) << 4));
  }
}
template <int SLOT_ID, int KS>
__device__ __forceinline__ void read_b(const unsigned char* smem, int wn, int fr, int fg, bf16x8 (&b)[2]) {
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int row = wn * 32 + g * 16 + fr;
    b[g] = *reinterpret_cast<const bf16x8*>(smem + SLOT_ID * SLOT + row * (BK * 2) + (((fg + 4 * KS) ^ swz(row)) << 4));
  }
}

// 8 MFMAs: one k-step of a 64x32 quadrant.  The W fragment is the MFMA's first operand, so the accumulator holds C^T: lane
// (fr, fg) of acc[f][g] owns row f*16 + fr and the FOUR CONSECUTIVE columns g*16 + fg*4 + [0,4) - one 8-byte store per
// accumulator in the epilogue instead of four 2-byte ones (the LDS store issue of the epilogue was ~5 % of a K = 5120 GEMM).
__device__ __forceinline__ void mma_half(f32x4 (&acc)[4][2], const bf16x8 (&a)[4], const bf16x8 (&b)[2]) {
  __builtin_amdgcn_s_setprio(1);
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int g = 0; g < 2; ++g) acc[f][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[g], a[f], acc[f][g], 0, 0, 0);
  __builtin_amdgcn_s_setprio(0);
}

#define CE_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0)
#define CE_VM(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#define CE_BAR() __builtin_amdgcn_s_barrier()

__device__ __forceinline__ void tile_origin(int wg, int tiles_m, int tiles_n, int& m0, int& n0) {
  constexpr int GROUP = 4;
  const int group_sz = GROUP * tiles_n;
  const int gid = wg / group_sz;
  const int first_m = gid * GROUP;
  const int gm = min(tiles_m - first_m, GROUP);
  const int tm = first_m + (wg % group_sz) % gm;
  const int tn = (wg % group_sz) / gm;
  m0 = tm * BM;
  n0 = tn * BN;
}

// Epilogue shared by the GEMM kernel and the split-K reduce kernel: the accumulators go through LDS in two passes
// of 128 tile rows so that the global stores (and the residual reads) are 16-byte row-contiguous.
template <int EPI>
__device__ __forceinline__ void epilogue256(const f32x4 (&acc)[2][2][4][2], unsigned char* smem, int tid, int wm, int wn,
                                            int fr, int fg, int m0, int n0, bf16* __restrict__ C,
                                            const float* __restrict__ bias, const float* __restrict__ gate,
                                            const bf16* __restrict__ res, int M, int N, int ldc, int ldres, int gate_rows) {
  // column bias of this lane's four (j, g) column groups, the same for both passes: four 16-B loads issued together, on clamped
  // addresses and without a per-lane guard (a guarded load compiles to its own branch with a vmcnt(0) behind it - seven serial
  // round trips per tile before this was hoisted); n is a multiple of 4 and N of 8, so "n < N" covers all four columns
  f32x4 bcol[2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int n = n0 + j * 128 + wn * 32 + g * 16 + fg * 4;
      bcol[j][g] = (EPI != EPI_BIAS_ROW && bias != nullptr) ? *reinterpret_cast<const f32x4*>(bias + min(n, N - 4)) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  // ---- epilogue, two passes of 128 tile rows (i = 0, 1)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (i == 1) __syncthreads();
    float brow[4] = {0.f, 0.f, 0.f, 0.f};  // EPI_BIAS_ROW: the bias of this lane's four rows (swapped operand roles)
    if (EPI == EPI_BIAS_ROW) {
#pragma unroll
      for (int f = 0; f < 4; ++f) brow[f] = bias[min(m0 + i * 128 + wm * 64 + f * 16 + fr, M - 1)];
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int cl = j * 128 + wn * 32 + g * 16 + fg * 4;
        f32x4 bv = bcol[j][g];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          const int rl = wm * 64 + f * 16 + fr;
          const f32x4 v = acc[i][j][f][g];
          if (EPI == EPI_BIAS_ROW) bv[0] = bv[1] = bv[2] = bv[3] = brow[f];
          const u32x2 pk = {pack_bf16(v[0] + bv[0], v[1] + bv[1]), pack_bf16(v[2] + bv[2], v[3] + bv[3])};
          *reinterpret_cast<u32x2*>(smem + rl * CROW + cl * 2) = pk;
        }
      }
    __syncthreads();
    epi_chunks<EPI, 8>(smem, CROW, [&](int tt, int& rl, int& cc, int& mr) { const int c = tid + 512 * tt; rl = mr = c >> 5; cc = c & 31; },
                       m0 + i * 128, n0, C, gate, res, M, N, ldc, ldres, gate_rows);
  }
}

template <int EPI, bool STAG>
__global__ __launch_bounds__(512) void gemm_bf16_256(const bf16* __restrict__ A, const bf16* __restrict__ W,
                                                     bf16* __restrict__ C, const float* __restrict__ bias,
                                                     const float* __restrict__ gate, const bf16* __restrict__ res, int M,
                                                     int N, int K, int lda, int ldw, int ldc, int ldres, int gate_rows,
                                                     int tiles_m, int tiles_n, int t_full, int split,
                                                     float* __restrict__ ws, uint32_t a_seg_magic, uint32_t a_seg_extra,
                                                     uint32_t w_seg_magic, uint32_t w_seg_extra) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int fr = lane & 15, fg = lane >> 4;

  // Blocks [0, t_full) own whole tiles (t_full is a multiple of the CU count, or every tile when nothing is split).
  // The tail tiles that would otherwise run as a partially filled last round are cut `split` ways along K; each
  // tail block leaves an fp32 slab in `ws` and gemm256_reduce sums the slabs and applies the epilogue.
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
  tile_origin(wg, tiles_m, tiles_n, m0, n0);

  Stager st;
  st.a_base = reinterpret_cast<const char*>(A);
  st.w_base = reinterpret_cast<const char*>(W);
  st.kt0 = kt0;
  st.a_seg_magic = a_seg_magic;
  st.a_seg_extra = a_seg_extra;
  st.w_seg_magic = w_seg_magic;
  st.w_seg_extra = w_seg_extra;
  st.wave = wave;
  st.kt_last = ktn - 1;
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int row = (r * 8 + wave) * 8 + (lane >> 3);        // row inside the 128-row half-tile
      const int chunk = (lane & 7) ^ swz(row);                 // logical chunk that must land at position lane&7
      st.a_off[h][r] = (uint32_t)min(m0 + h * 128 + row, M - 1) * (uint32_t)(lda * 2) + chunk * 16;
      st.w_off[h][r] = (uint32_t)min(n0 + h * 128 + row, N - 1) * (uint32_t)(ldw * 2) + chunk * 16;
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

  // prologue: even tile 0 (4 halves) + odd tile 1 (A0, B0, B1); O.A1 follows in phase 1
  stage_half<S_A0>(smem, st, 0);
  stage_half<S_B0>(smem, st, 0);
  stage_half<S_B1>(smem, st, 0);
  stage_half<S_A1>(smem, st, 0);
  stage_half<4 + S_A0>(smem, st, 1);
  stage_half<4 + S_B0>(smem, st, 1);
  stage_half<4 + S_B1>(smem, st, 1);
  CE_VM(4);  // everything but the last two half-tiles (O.B0, O.B1) has landed
  CE_BAR();

  if constexpr (!STAG) {
  // Operand registers, software-pipelined across phases: a register set is refilled (ds_read) right after the
  // last MFMA that consumed it, so the next phase's fragments are already in flight when its barrier opens.
  bf16x8 ra0[4], ra1[4], b0k0[2], b0k1[2], b1k0[2], b1k1[2];  // A-sub (k-step 0/1), W-sub0, W-sub1
  read_a<S_A0, 0>(smem, wm, fr, fg, ra0);
  read_a<S_A0, 1>(smem, wm, fr, fg, ra1);
  read_b<S_B0, 0>(smem, wn, fr, fg, b0k0);
  read_b<S_B0, 1>(smem, wn, fr, fg, b0k1);
  read_b<S_B1, 0>(smem, wn, fr, fg, b1k0);
  read_b<S_B1, 1>(smem, wn, fr, fg, b1k1);

  // (the ablation build of round 1 - this loop with its barriers / staging / fragment reads / MFMAs removed one at a time - gave
  // the cost table of DESIGN.md section 4.1, profiles/r01_gemm_ablate_14400x15360x5120.txt; git history has the switches)
#define CE_STAGE(SLOT_EVEN, SLOT_ODD, CURV, TILEV) \
  if (CURV == 0) stage_half<SLOT_EVEN>(smem, st, (TILEV)); else stage_half<SLOT_ODD>(smem, st, (TILEV));
#define CE_LBAR() CE_BAR()
#define CE_RDA(S, K, R) read_a<S, K>(smem, wm, fr, fg, R)
#define CE_RDB(S, K, R) read_b<S, K>(smem, wn, fr, fg, R)
#define CE_MMA(ACC, A, B) mma_half(ACC, A, B)

  // The LDS-DMA issue (address math + M0 + 2 global_load_lds) sits BETWEEN the two 8-MFMA k-steps of a phase, so it
  // overlaps the matrix pipe instead of extending the barrier-to-first-MFMA gap.
#define CE_TILE_PHASES(CUR, NXT, TILE)                                                                      \
  /* phase 1: Q00 */                                                                                        \
  CE_LBAR();                                                                                                 \
  CE_LGKM0();                                                                                               \
  CE_MMA(acc[0][0], ra0, b0k0);                                                                           \
  CE_STAGE(4 + S_A1, S_A1, CUR, (TILE) + 1)                                                                 \
  CE_MMA(acc[0][0], ra1, b0k1);                                                                           \
  /* phase 2: Q01, refill the A registers with A-sub1 as they drain; A-sub1 of this tile (staged in phase 1 of   \
     the previous tile, eight DMA instructions ago) is first read here, so its counted wait sits here too */     \
  CE_VM(8);                                                                                                 \
  CE_LBAR();                                                                                                 \
  CE_MMA(acc[0][1], ra0, b1k0);                                                                           \
  CE_RDA(CUR * 4 + S_A1, 0, ra0);                                                         \
  CE_STAGE(S_A0, 4 + S_A0, CUR, (TILE) + 2)                                                                 \
  CE_MMA(acc[0][1], ra1, b1k1);                                                                           \
  CE_RDA(CUR * 4 + S_A1, 1, ra1);                                                         \
  /* phase 3: Q11 */                                                                                        \
  CE_LBAR();                                                                                                 \
  CE_LGKM0();                                                                                               \
  CE_MMA(acc[1][1], ra0, b1k0);                                                                           \
  CE_STAGE(S_B0, 4 + S_B0, CUR, (TILE) + 2)                                                                 \
  CE_MMA(acc[1][1], ra1, b1k1);                                                                           \
  /* phase 4: Q10; the counted wait retires the NEXT tile's slots, whose fragments are prefetched here */   \
  CE_VM(6);                                                                                                 \
  CE_LBAR();                                                                                                 \
  CE_RDB(NXT * 4 + S_B1, 0, b1k0);                                                        \
  CE_RDB(NXT * 4 + S_B1, 1, b1k1);                                                        \
  CE_MMA(acc[1][0], ra0, b0k0);                                                                           \
  CE_RDA(NXT * 4 + S_A0, 0, ra0);                                                         \
  CE_RDB(NXT * 4 + S_B0, 0, b0k0);                                                        \
  CE_STAGE(S_B1, 4 + S_B1, CUR, (TILE) + 2)                                                                 \
  CE_MMA(acc[1][0], ra1, b0k1);                                                                           \
  CE_RDA(NXT * 4 + S_A0, 1, ra1);                                                         \
  CE_RDB(NXT * 4 + S_B0, 1, b0k1);

  const int npairs = ktn >> 1;
  for (int it = 0; it < npairs; ++it) {
    const int t = 2 * it;
    CE_TILE_PHASES(0, 1, t)      // even K-tile t   (slots 0-3); stages O.A1(t+1), E.A0/E.B0/E.B1(t+2)
    CE_TILE_PHASES(1, 0, t + 1)  // odd  K-tile t+1 (slots 4-7); stages E.A1(t+2), O.A0/O.B0/O.B1(t+3)
  }
#undef CE_TILE_PHASES
#undef CE_STAGE
  } else {
    // ---- staggered form (cdna guide 8-phase template): every phase is a LOAD segment (this phase's fragment reads +
    // one LDS-DMA stage) and a pure 16-MFMA segment, each closed by an s_barrier; the wave group wm = 1 runs one barrier
    // behind wm = 0, so on every SIMD one wave multiplies while its partner loads.
    //     phase of tile T     1                 2               3               4
    //     reads (LOAD seg.)   A0(T), B0(T)      B1(T)           A1(T)           -
    //     stage (LOAD seg.)   other.A1(T+1)     cur.A0(T+2)     cur.B0(T+2)     cur.B1(T+2), then vmcnt(6)
    //     MFMA segment        Q00               Q01             Q11             Q10
    // Hazards with the one-barrier lag: a slot is restaged >= 1 phase after the phase that issued its last reads, and
    // read >= 1 phase after the phase whose LOAD segment ends with the counted vmcnt that retires it (three half-tiles
    // stay in flight across the wait).
    bf16x8 ra0[4], ra1[4], b0k0[2], b0k1[2], b1k0[2], b1k1[2];
    if (wm == 1) CE_BAR();
#define CE_STAGE(SLOT_EVEN, SLOT_ODD, CURV, TILEV) \
  if (CURV == 0) stage_half<SLOT_EVEN>(smem, st, (TILEV)); else stage_half<SLOT_ODD>(smem, st, (TILEV));
#define CE_TILE_STAG(CUR, TILE)                                                                              \
  /* phase 1 */                                                                                              \
  read_a<CUR * 4 + S_A0, 0>(smem, wm, fr, fg, ra0);                                                          \
  read_a<CUR * 4 + S_A0, 1>(smem, wm, fr, fg, ra1);                                                          \
  read_b<CUR * 4 + S_B0, 0>(smem, wn, fr, fg, b0k0);                                                         \
  read_b<CUR * 4 + S_B0, 1>(smem, wn, fr, fg, b0k1);                                                         \
  CE_STAGE(4 + S_A1, S_A1, CUR, (TILE) + 1)                                                                  \
  CE_BAR();                                                                                                  \
  CE_LGKM0();                                                                                                \
  mma_half(acc[0][0], ra0, b0k0);                                                                            \
  mma_half(acc[0][0], ra1, b0k1);                                                                            \
  CE_BAR();                                                                                                  \
  /* phase 2 */                                                                                              \
  read_b<CUR * 4 + S_B1, 0>(smem, wn, fr, fg, b1k0);                                                         \
  read_b<CUR * 4 + S_B1, 1>(smem, wn, fr, fg, b1k1);                                                         \
  CE_STAGE(S_A0, 4 + S_A0, CUR, (TILE) + 2)                                                                  \
  CE_BAR();                                                                                                  \
  CE_LGKM0();                                                                                                \
  mma_half(acc[0][1], ra0, b1k0);                                                                            \
  mma_half(acc[0][1], ra1, b1k1);                                                                            \
  CE_BAR();                                                                                                  \
  /* phase 3 */                                                                                              \
  read_a<CUR * 4 + S_A1, 0>(smem, wm, fr, fg, ra0);                                                          \
  read_a<CUR * 4 + S_A1, 1>(smem, wm, fr, fg, ra1);                                                          \
  CE_STAGE(S_B0, 4 + S_B0, CUR, (TILE) + 2)                                                                  \
  CE_BAR();                                                                                                  \
  CE_LGKM0();                                                                                                \
  mma_half(acc[1][1], ra0, b1k0);                                                                            \
  mma_half(acc[1][1], ra1, b1k1);                                                                            \
  CE_BAR();                                                                                                  \
  /* phase 4 */                                                                                              \
  CE_STAGE(S_B1, 4 + S_B1, CUR, (TILE) + 2)                                                                  \
  CE_VM(6);                                                                                                  \
  CE_BAR();                                                                                                  \
  mma_half(acc[1][0], ra0, b0k0);                                                                            \
  mma_half(acc[1][0], ra1, b0k1);                                                                            \
  CE_BAR();

    const int npairs = ktn >> 1;
    for (int it = 0; it < npairs; ++it) {
      const int t = 2 * it;
      CE_TILE_STAG(0, t)
      CE_TILE_STAG(1, t + 1)
    }
#undef CE_TILE_STAG
#undef CE_STAGE
    if (wm == 0) CE_BAR();
  }
  CE_VM(0);  // surplus prefetches (LDS-DMA and fragment reads) must retire before the epilogue reuses the LDS
  CE_LGKM0();
  CE_BAR();

  if (partial) {
    float* slab = ws + (size_t)(blockIdx.x - t_full) * (BM * BN);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
          for (int g = 0; g < 2; ++g)
            *reinterpret_cast<f32x4*>(slab + ((((i * 2 + j) * 4 + f) * 2 + g) * 512 + tid) * 4) = acc[i][j][f][g];
    return;
  }
  epilogue256<EPI>(acc, smem, tid, wm, wn, fr, fg, m0, n0, C, bias, gate, res, M, N, ldc, ldres, gate_rows);
}

// Sums the `split` fp32 slabs of one QUADRANT (128 x 128) of a tail tile - same thread <-> accumulator mapping as the GEMM
// kernel - and applies the epilogue.  grid = 4 x the number of tail tiles: with one block per tile the reduce ran on as
// few as six CUs (FFN-up: six tail tiles, eight slabs each, 68 us) and added up to 2.3 % of a denoising step.
constexpr int QROW = 128 * 2 + 16;  // padded staging row of a quadrant (272 B)
template <int EPI>
__global__ __launch_bounds__(512) void gemm256_reduce(bf16* __restrict__ C, const float* __restrict__ bias,
                                                      const float* __restrict__ gate, const bf16* __restrict__ res, int M,
                                                      int N, int ldc, int ldres, int gate_rows, int tiles_m, int tiles_n,
                                                      int t_full, int split, const float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int fr = lane & 15, fg = lane >> 4;
  const int tile = blockIdx.x >> 2, qi = (blockIdx.x >> 1) & 1, qj = blockIdx.x & 1;
  int m0, n0;
  tile_origin(t_full + tile, tiles_m, tiles_n, m0, n0);
  m0 += qi * 128;
  n0 += qj * 128;
  const float* slab = ws + (size_t)tile * split * (BM * BN);
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int cl = wn * 32 + g * 16 + fg * 4;
    const int n = n0 + cl;
    f32x4 bv = (EPI != EPI_BIAS_ROW && bias != nullptr) ? *reinterpret_cast<const f32x4*>(bias + min(n, N - 4)) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      if (EPI == EPI_BIAS_ROW) bv[0] = bv[1] = bv[2] = bv[3] = bias[min(m0 + wm * 64 + f * 16 + fr, M - 1)];
      const int e = (((((qi * 2 + qj) * 4 + f) * 2 + g) * 512) + tid) * 4;
      f32x4 v = *reinterpret_cast<const f32x4*>(slab + e);
      for (int sidx = 1; sidx < split; ++sidx) v += *reinterpret_cast<const f32x4*>(slab + (size_t)sidx * (BM * BN) + e);
      const int rl = wm * 64 + f * 16 + fr;
      const u32x2 pk = {pack_bf16(v[0] + bv[0], v[1] + bv[1]), pack_bf16(v[2] + bv[2], v[3] + bv[3])};
      *reinterpret_cast<u32x2*>(smem + rl * QROW + cl * 2) = pk;
    }
  }
  __syncthreads();
  epi_chunks<EPI, 4>(smem, QROW, [&](int tt, int& rl, int& cc, int& mr) { const int c = tid + 512 * tt; rl = mr = c >> 4; cc = c & 15; }, m0, n0, C,
                     gate, res, M, N, ldc, ldres, gate_rows);
}

}  // namespace

// returns 1 when the 256-tile kernel can take the shape
extern "C" int ce_gemm256_supported(int M, int N, int K, int lda, int ldw) {
  const int kt = K / BK;
  if ((K % BK) || (kt & 1) || kt < 2) return 0;
  if ((long long)M * lda * 2 >= (1ll << 32) || (long long)N * ldw * 2 >= (1ll << 32)) return 0;  // 32-bit DMA offsets
  return 1;
}

#ifdef CE_DIAGNOSTICS
static bool staggered = false;
extern "C" void ce_gemm256_set_staggered(int on) { staggered = on != 0; }
#else
static constexpr bool staggered = false;
#endif

// Scratch for the split-K tail (fp32 slabs); without it every tile runs whole.  Host-side registry, not on the data path: one DEFAULT
// scratch per device (ce_set_gemm_workspace, registered for the current device) and, for callers that run GEMMs on several streams of
// one device at once, one scratch per (device, stream) (ce_set_gemm_workspace_stream) - a launch uses its stream's own scratch when one
// is registered and the device default otherwise, so two streams never share slabs unless the caller registered nothing for them.
namespace {
struct WsSlot {
  float* ptr = nullptr;
  size_t bytes = 0;
};
struct WsStream {
  int dev = -1;
  hipStream_t stream = nullptr;
  WsSlot ws;
};
constexpr int WS_STREAMS = 32;
WsSlot g_ws_dev[CE_MAX_DEVICES];
WsStream g_ws_stream[WS_STREAMS];
int g_ws_stream_next = 0;
int g_cus_dev[CE_MAX_DEVICES] = {};
std::mutex g_ws_mutex;
int device_cus(int slot) {
  if (g_cus_dev[slot] == 0) {
    int dev = 0, cus = 0;
    g_cus_dev[slot] = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) ? cus : 256;
  }
  return g_cus_dev[slot];
}
}  // namespace

CE_API int ce_set_gemm_workspace(void* ptr, size_t bytes) {
  std::lock_guard<std::mutex> lock(g_ws_mutex);
  const int slot = ce_device_slot();
  g_ws_dev[slot].ptr = reinterpret_cast<float*>(ptr);
  g_ws_dev[slot].bytes = ptr ? bytes : 0;
  (void)device_cus(slot);
  return CE_OK;
}

CE_API int ce_set_gemm_workspace_stream(hipStream_t stream, void* ptr, size_t bytes) {
  std::lock_guard<std::mutex> lock(g_ws_mutex);
  const int slot = ce_device_slot();
  int at = -1;
  for (int i = 0; i < WS_STREAMS; ++i)
    if (g_ws_stream[i].dev == slot && g_ws_stream[i].stream == stream) at = i;
  if (ptr == nullptr) {  // unregister: the stream falls back to the device default
    if (at >= 0) g_ws_stream[at] = WsStream{};
    return CE_OK;
  }
  if (at < 0) {
    for (int i = 0; i < WS_STREAMS && at < 0; ++i)
      if (g_ws_stream[i].dev < 0) at = i;
    if (at < 0) at = g_ws_stream_next++ % WS_STREAMS;  // full: the oldest registration makes room (its stream uses the default again)
  }
  g_ws_stream[at].dev = slot;
  g_ws_stream[at].stream = stream;
  g_ws_stream[at].ws = WsSlot{reinterpret_cast<float*>(ptr), bytes};
  (void)device_cus(slot);
  return CE_OK;
}

// the scratch a launch on `stream` (current device) may use - for every large-tile main loop (this file, ce_gemm256w4.hip, ce_gemm384.hip,
// ce_gemm_fp8w4.hip) and for the tile choice of ce_gemm.hip
extern "C" void ce_gemm256_workspace(hipStream_t stream, float** ws, size_t* bytes, int* cus) {
  std::lock_guard<std::mutex> lock(g_ws_mutex);
  const int slot = ce_device_slot();
  WsSlot w = g_ws_dev[slot];
  for (int i = 0; i < WS_STREAMS; ++i)
    if (g_ws_stream[i].dev == slot && g_ws_stream[i].stream == stream) w = g_ws_stream[i].ws;
  *ws = w.ptr;
  *bytes = w.bytes;
  *cus = device_cus(slot);
}

extern "C" int ce_gemm256_launch(const void* A, const void* W, void* C, const float* bias, int epilogue, const float* gate,
                                 const void* res, int M, int N, int K, int lda, int ldw, int ldc, int ldres, int gate_rows,
                                 int a_seg_k, long long a_seg_stride, int w_seg_k, long long w_seg_stride, hipStream_t stream) {
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const int nwg = tiles_m * tiles_n, kt = K / BK;
  // segmented operand: column k lives at base + (k / seg_k) * seg_stride + row * ld + k % seg_k (elements)
  uint32_t a_seg_magic = 0, a_seg_extra = 0, w_seg_magic = 0, w_seg_extra = 0;
  auto seg = [&](int seg_k, long long seg_stride, uint32_t& magic, uint32_t& extra_out) -> int {
    if (seg_k <= 0 || seg_k >= K) return CE_OK;
    if (seg_k % BK) return CE_ERR_SHAPE;
    const int tps = seg_k / BK;
    magic = 65536u / (uint32_t)tps + 1u;
    for (int t = 0; t < kt; ++t)
      if ((int)(((uint32_t)t * magic) >> 16) != t / tps) return CE_ERR_SHAPE;
    const long long extra = (seg_stride - seg_k) * 2;  // bytes on top of the contiguous advance
    if (extra < 0 || extra * (K / seg_k) >= (1ll << 32)) return CE_ERR_SHAPE;
    extra_out = (uint32_t)extra;
    return CE_OK;
  };
  if (int rc = seg(a_seg_k, a_seg_stride, a_seg_magic, a_seg_extra)) return rc;
  if (int rc = seg(w_seg_k, w_seg_stride, w_seg_magic, w_seg_extra)) return rc;
  // split-K only for the tail of the last, partially filled round of workgroups
  float* g_ws = nullptr;
  size_t g_ws_bytes = 0;
  int g_cus = 256;
  ce_gemm256_workspace(stream, &g_ws, &g_ws_bytes, &g_cus);
  int tail = nwg % g_cus, split = 1;
  if (tail > 0 && g_ws != nullptr) {
    for (int s = std::min(g_cus / tail, 8); s >= 2; --s)
      if (kt % (2 * s) == 0 && (size_t)tail * s * BM * BN * sizeof(float) <= g_ws_bytes) {
        split = s;
        break;
      }
  }
  if (split == 1) tail = 0;
  const int t_full2 = nwg - tail;
  dim3 grid(t_full2 + tail * split), block(512);
  static bool attr_done_[CE_MAX_DEVICES][8] = {};
  bool* attr_done = attr_done_[ce_device_slot()];
#define CE_LAUNCH(E)                                                                                                  \
  do {                                                                                                                \
    if (!attr_done[E]) {                                                                                              \
      (void)hipFuncSetAttribute((const void*)gemm_bf16_256<E, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES); \
      (void)hipFuncSetAttribute((const void*)gemm_bf16_256<E, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);  \
      (void)hipFuncSetAttribute((const void*)gemm256_reduce<E>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * QROW);     \
      attr_done[E] = true;                                                                                            \
    }                                                                                                                 \
    if (staggered)                                                                                                    \
      hipLaunchKernelGGL((gemm_bf16_256<E, true>), grid, block, LDS_BYTES, stream, (const bf16*)A, (const bf16*)W, (bf16*)C, \
                         bias, gate, (const bf16*)res, M, N, K, lda, ldw, ldc, ldres, gate_rows, tiles_m, tiles_n,    \
                         t_full2, split, g_ws, a_seg_magic, a_seg_extra, w_seg_magic, w_seg_extra);                                             \
    else                                                                                                              \
      hipLaunchKernelGGL((gemm_bf16_256<E, false>), grid, block, LDS_BYTES, stream, (const bf16*)A, (const bf16*)W, (bf16*)C, \
                         bias, gate, (const bf16*)res, M, N, K, lda, ldw, ldc, ldres, gate_rows, tiles_m, tiles_n,    \
                         t_full2, split, g_ws, a_seg_magic, a_seg_extra, w_seg_magic, w_seg_extra);                                             \
    if (tail)                                                                                                         \
      hipLaunchKernelGGL((gemm256_reduce<E>), dim3(4 * tail), block, 128 * QROW, stream, (bf16*)C, bias, gate,        \
                         (const bf16*)res, M, N, ldc, ldres, gate_rows, tiles_m, tiles_n, t_full2, split, g_ws);      \
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
