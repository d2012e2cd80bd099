// Self-attention forward on v_mfma_f32_16x16x32_bf16 (round 5): the 16 x 16 x 32 geometry for the V^T / LDS-DMA form of the DiT's self-attention
// (ce_attention_vt_bf16, plain row layout), selected with ce_set_attention_waves(16).  Same contract as attn_fwd_sp_kernel<false, true> in
// ce_attn.hip: O = softmax(Q K^T / sqrt(128)) V per head, head_dim 128, bf16 in / out, fp32 accumulation, Q pre-multiplied by
// softmax_scale * log2(e) and rounded to bf16 once (scores leave the matrix pipe in the exp2 domain), P rounded to bf16 for the second product,
// row sums in fp32 of the un-rounded P.  Replaces F.scaled_dot_product_attention at transformer_chronoedit.py:91-104.
//
// Why a second geometry, and what it measured (profiles/r05_attention_16x16x32.txt): the bare MFMA streams (profiles/r01_mfma_rate_probe.txt) and a
// synthetic tile loop with this kernel's filler load (tools/probes/attn_shape_probe.hip) run 6 ... 8 % faster on 16x16x32 than on 32x32x16 - but only
// at FOUR waves per SIMD.  At the two waves per SIMD the register budget of a head_dim-128 kernel allows (O^T 64 + Q 32 + S 32 registers per lane) the
// geometries are level, and this compiler-scheduled body runs 0.93 ... 1.05 PFLOP/s against the hand-scheduled 32x32x16 body's 1.21 ... 1.28: it is
// an OPT-IN (ce_set_attention_waves(16)), parity-tested record of that result, not the production path.
//
//  * workgroup = 8 waves x 32 query rows = 256 query rows of one head; 64-key tiles; three stages of [K 16 KiB | V^T 16 KiB] filled by LDS-DMA
//    (global_load_lds, 16 B per lane, source-side chunk swizzles) two tiles ahead, one barrier per tile.
//  * a wave alternates ONE matrix block per tile - O += V^T(t).P^T(t), the row sums, S(t+1) = K(t+1).Q^T: 68 MFMAs back to back - with ONE vector
//    block (mask, row maxima, exponentials, packing); the two waves of a SIMD run the two in antiphase.
//  * S^T = K.Q^T: first operand = K fragment [16 keys x 32 d] (one ds_read_b128 per lane: chunk (4 ks + g) ^ (key & 15) of the key's 256-byte
//    row), second = Q fragment [32 d x 16 queries] held in registers (2 query blocks x 4 k-steps).  Lane (n, g) of accumulator s[kb][qb] owns
//    query 16 qb + n and LDS rows 16 kb + 4 g + j of the tile: a query's statistics live in 4 lanes (n, n + 16, n + 32, n + 48).
//  * O^T += V^T.P^T: the P operand comes straight out of the S registers (k-slots 8 g .. 8 g + 7 of k-step h = this lane's values of
//    s[2 h][.] and s[2 h + 1][.]).  The K ROWS of a tile are placed in LDS in the order that makes those eight slots eight CONSECUTIVE keys -
//    LDS row 16 kb + 4 g + j holds key 32 (kb >> 1) + 8 g + 4 (kb & 1) + j; the DMA's per-lane source row does the permutation for nothing -
//    so the V^T fragment [16 d x 32 keys] of a lane is ONE 16-byte read of natural-order V^T (unit 4 h + g of the channel's 128-byte row, in
//    slot (4 h + g) ^ ((row >> 1) & 7): conflict-free reads, a whole-16-byte permutation on the DMA's source side).
//  * online softmax with a LAZY offset: the S accumulators START at minus the row's offset (scores arrive as "score - offset": no subtract),
//    the offset is the exact maximum of tile 0 and moves afterwards only when a tile's maximum exceeds it by more than 2^8 (P stays below 2^8
//    in fp32 / bf16: no precision is lost by an offset that lags), so accumulators are rescaled on the first tiles only; the row sums of the
//    rounded P come off the matrix pipe (an all-ones operand against P: 4 MFMAs per tile instead of 32 adds per lane).
//  * work order: attn_fwd_sp_kernel's (an XCD keeps its heads; batch folded into the item index; persistent workgroups).
//  * PRECONDITION on V^T: the padded columns [len, ceil64(len)) of every channel row must hold FINITE values (the callers zero-fill the buffer;
//    ce_v_transpose_bf16 and the CE_EPI_BIAS_ROW projection write zeros there): masked keys get P = 0, and 0 x NaN garbage would still be NaN.
//  * Built into libchronoedit_hip_diag.so ONLY (round 6): the product library neither contains nor dispatches to this body.
#include "ce_common.h"

namespace {

constexpr int HD = 128, QW = 32, KVB = 64;
constexpr int K_TILE = KVB * HD * 2;   // 16 KiB: 64 key rows of 256 B
constexpr int V_TILE = HD * KVB * 2;   // 16 KiB: 128 channel rows of 128 B
constexpr int STAGE = K_TILE + V_TILE;
constexpr int NST = 3;
constexpr float NEG_BIG = -1.0e30f;
constexpr float LAZY = 8.0f;  // exp2 domain: the running offset moves when a tile maximum exceeds it by more than this

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

__global__ __launch_bounds__(512, 2) void attn_fwd_x16_kernel(const bf16* __restrict__ Q_, const bf16* __restrict__ K_, const bf16* __restrict__ Vt_,
                                                              bf16* __restrict__ O_, int Nq, int Nkv, int H, int ldq, int ldk, int ldvt, int ldo,
                                                              int nqb, float sl2, int batch, int vt_cols) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  const int ntiles = (Nkv + KVB - 1) / KVB;
#pragma clang loop unroll(disable)
  for (int item = blockIdx.x; item < nqb * H * batch; item += gridDim.x) {
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
    const bf16* Q = Q_ + (size_t)bz * Nq * ldq;
    const bf16* K = K_ + (size_t)bz * Nkv * ldk;
    const bf16* Vt = Vt_ + (size_t)bz * vt_cols;  // sample b's keys: columns [b vt_cols, ...)
    bf16* O = O_ + (size_t)bz * Nq * ldo;
    const int hoff = head * HD;
    const int q0 = qb * (QW * 8) + wave * QW;
    const bool active = q0 < Nq;  // wave-uniform: a wave past the last query row only stages tiles and keeps the barriers

    // Q fragments, pre-scaled: qf[qblk][ks] = Q[q0 + 16 qblk + n][32 ks + 8 g .. + 8] * sl2, rounded to bf16
    bf16x8 qf[2][4];
#pragma unroll
    for (int qblk = 0; qblk < 2; ++qblk) {
      const bf16* qrow = Q + (size_t)min(q0 + 16 * qblk + n, Nq - 1) * ldq + hoff + 8 * g;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const u32x4 raw = *reinterpret_cast<const u32x4*>(qrow + 32 * ks);
        u32x4 sc;
#pragma unroll
        for (int w = 0; w < 4; ++w) sc[w] = pack_bf16(bf16lo(raw[w]) * sl2, bf16hi(raw[w]) * sl2);
        qf[qblk][ks] = __builtin_bit_cast(bf16x8, sc);
      }
    }

    // LDS-DMA sources.  K tile: piece p of wave w = LDS rows 8 w + 4 p + (lane >> 4) <- key key_of(row), 16-byte slot lane & 15 <- source chunk
    // slot ^ (row & 15).
    // V^T tile: piece p of wave w = channel rows 16 w + 8 p + (lane >> 3), 16-byte slot lane & 7 <- source unit slot ^ ((row >> 1) & 7).
    const int k_row0 = 8 * wave + (lane >> 4), k_slot = lane & 15;
    const int v_row0 = 16 * wave + (lane >> 3), v_slot = lane & 7;
    auto stage_tile = [&](int t, int s) __attribute__((always_inline)) {
      unsigned char* st = smem + s * STAGE;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int row = k_row0 + 4 * p;
        const int key = 32 * (row >> 5) + 8 * ((row >> 2) & 3) + 4 * ((row >> 4) & 1) + (row & 3);  // the key LDS row `row` holds (see the header)
        const int kr = min(t * KVB + key, Nkv - 1);
        __builtin_amdgcn_global_load_lds((gbl_void*)(K + (size_t)kr * ldk + hoff + ((k_slot ^ (row & 15)) << 3)), (lds_void*)(st + (2 * wave + p) * 1024), 16, 0,
                                         0);
      }
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int row = v_row0 + 8 * p;
        __builtin_amdgcn_global_load_lds((gbl_void*)(Vt + (size_t)(hoff + row) * ldvt + t * KVB + ((v_slot ^ ((row >> 1) & 7)) << 3)),
                                         (lds_void*)(st + K_TILE + (2 * wave + p) * 1024), 16, 0, 0);
      }
    };
    // fragment read offsets inside a stage
    //   K fragment (kb, ks): row 16 kb + n, chunk (4 ks + g) ^ n                       -> kb * 4096 + n * 256 + (((4 ks + g) ^ n) << 4)
    //   V^T fragment (h, d): row 16 d + n, 16-byte unit 4 h + g in slot (4 h + g) ^ ((n >> 1) & 7)   ((row >> 1) & 7 == (n >> 1) & 7)
    const int v_sw = (n >> 1) & 7;
    const int v_rd = K_TILE + n * 128;

    f32x4 o[8][2], lacc[2];
#pragma unroll
    for (int qblk = 0; qblk < 2; ++qblk) {
      lacc[qblk] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int d = 0; d < 8; ++d) o[d][qblk] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // all-ones A operand: ones.P^T leaves every lane the sum over the 32 key slots of a k-step for its query - the row sums of the ROUNDED P
    // come off the matrix pipe (4 MFMAs per tile) instead of 32 adds per lane, and need no cross-lane reduction at the end
    u32x4 ones_w = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    asm volatile("" : "+v"(ones_w));
    const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_w);
    float m_run[2] = {0.f, 0.f};  // offset (exp2 domain) the scores of a row are taken against; S accumulators START at -m_run
    float mx_cur[2] = {0.f, 0.f}; // maximum of s_cur (already relative to m_run), identical in the 4 lanes of a query

    // S^T(t) = K(t).Q^T - m_run, masked; returns the row maxima (relative to m_run)
    auto scores = [&](int t, f32x4 (&sx)[4][2], float (&mx)[2]) __attribute__((always_inline)) {
      const unsigned char* st = smem + (t % NST) * STAGE;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
        for (int qblk = 0; qblk < 2; ++qblk) sx[kb][qblk] = f32x4{-m_run[qblk], -m_run[qblk], -m_run[qblk], -m_run[qblk]};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(st + kb * 4096 + n * 256 + (((4 * ks + g) ^ n) << 4));
#pragma unroll
          for (int qblk = 0; qblk < 2; ++qblk) sx[kb][qblk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qblk][ks], sx[kb][qblk], 0, 0, 0);
        }
      }
    };
    auto mask_and_max = [&](int t, f32x4 (&sx)[4][2], float (&mx)[2]) __attribute__((always_inline)) {
      if ((t + 1) * KVB > Nkv) {  // key tail of the last tile: sx[kb][.][j] belongs to key 64 t + 32 (kb >> 1) + 8 g + 4 (kb & 1) + j
        const int base = t * KVB + 8 * g;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (base + 32 * (kb >> 1) + 4 * (kb & 1) + j >= Nkv) {
              sx[kb][0][j] = NEG_BIG;
              sx[kb][1][j] = NEG_BIG;
            }
      }
#pragma unroll
      for (int qblk = 0; qblk < 2; ++qblk) {
        float v = fmaxf(fmaxf(sx[0][qblk][0], sx[0][qblk][1]), fmaxf(sx[0][qblk][2], sx[0][qblk][3]));
#pragma unroll
        for (int kb = 1; kb < 4; ++kb) v = fmaxf(v, fmaxf(fmaxf(sx[kb][qblk][0], sx[kb][qblk][1]), fmaxf(sx[kb][qblk][2], sx[kb][qblk][3])));
        v = fmaxf(v, __shfl_xor(v, 16, 64));
        mx[qblk] = fmaxf(v, __shfl_xor(v, 32, 64));
      }
    };

    // prologue: tiles 0 and 1 on their way; S(0) against offset 0
    stage_tile(0, 0);
    stage_tile(min(1, ntiles - 1), 1);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    f32x4 s_cur[4][2];
    if (active) {
      scores(0, s_cur, mx_cur);
      mask_and_max(0, s_cur, mx_cur);
    }
    for (int t = 0; t < ntiles; ++t) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of tile t+1 have landed (nothing else of its is in flight)
      __builtin_amdgcn_s_barrier();                     // ... everybody's have; everybody is through iteration t-1: stage (t+2) % 3 is free
      stage_tile(min(t + 2, ntiles - 1), (t + 2) % NST);
      if (!active) continue;
      // ---- the offset of a row moves when its tile maximum outgrows it by more than 2^LAZY (tile 0: always, to the exact maximum)
#pragma unroll
      for (int qblk = 0; qblk < 2; ++qblk) {
        if (t == 0 || __any(mx_cur[qblk] > LAZY)) {
          const float delta = t == 0 ? mx_cur[qblk] : fmaxf(mx_cur[qblk], 0.f);
          m_run[qblk] += delta;
          if (t > 0) {
            const float alpha = __builtin_amdgcn_exp2f(-delta);
            lacc[qblk] *= alpha;
#pragma unroll
            for (int d = 0; d < 8; ++d) o[d][qblk] *= alpha;
          }
#pragma unroll
          for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int j = 0; j < 4; ++j) s_cur[kb][qblk][j] -= delta;
        }
      }
      // ---- VALU block: P(t) = exp2(s_cur) packed to bf16 (the other wave of the SIMD is in its matrix block meanwhile)
      u32x4 pw[2][2];  // [query block][key half]: 8 bf16 = the P operand of one k-step
#pragma unroll
      for (int qblk = 0; qblk < 2; ++qblk)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float p[2][4];
#pragma unroll
          for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int j = 0; j < 4; ++j) p[e][j] = __builtin_amdgcn_exp2f(s_cur[2 * h + e][qblk][j]);
          pw[qblk][h] = u32x4{pack_bf16(p[0][0], p[0][1]), pack_bf16(p[0][2], p[0][3]), pack_bf16(p[1][0], p[1][1]), pack_bf16(p[1][2], p[1][3])};
        }
      // ---- matrix block: O^T += V^T(t).P^T(t), the row sums, then S(t+1) into the registers P(t) came from - 68 MFMAs back to back
      {
        const unsigned char* st = smem + (t % NST) * STAGE;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
          for (int d = 0; d < 8; ++d) {
            const bf16x8 vf = *reinterpret_cast<const bf16x8*>(st + v_rd + d * 2048 + (((4 * h + g) ^ v_sw) << 4));
#pragma unroll
            for (int qblk = 0; qblk < 2; ++qblk)
              o[d][qblk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, __builtin_bit_cast(bf16x8, pw[qblk][h]), o[d][qblk], 0, 0, 0);
          }
#pragma unroll
          for (int qblk = 0; qblk < 2; ++qblk)
            lacc[qblk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, __builtin_bit_cast(bf16x8, pw[qblk][h]), lacc[qblk], 0, 0, 0);
        }
      }
      if (t + 1 < ntiles) {
        scores(t + 1, s_cur, mx_cur);
        mask_and_max(t + 1, s_cur, mx_cur);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the surplus prefetch must land before the next item stages over it
    // ---- normalise and store: lane (n, g) of o[d][qblk] owns query 16 qblk + n, channels 16 d + 4 g + [0, 4); every lane holds its row's sum
    if (active) {
#pragma unroll
      for (int qblk = 0; qblk < 2; ++qblk) {
        const float inv = 1.0f / lacc[qblk][0];
        const int q = q0 + 16 * qblk + n;
        if (q < Nq) {
          bf16* orow = O + (size_t)q * ldo + hoff + 4 * g;
#pragma unroll
          for (int d = 0; d < 8; ++d) {
            const u32x2 v = {pack_bf16(o[d][qblk][0] * inv, o[d][qblk][1] * inv), pack_bf16(o[d][qblk][2] * inv, o[d][qblk][3] * inv)};
            *reinterpret_cast<u32x2*>(orow + 16 * d) = v;
          }
        }
      }
    }
    __builtin_amdgcn_s_barrier();  // nobody stages the next item's tile 0 over a stage a slower wave still reads
  }
}

}  // namespace

// The plain-layout single-segment V^T attention on the 16 x 16 x 32 geometry (called by ce_attention_vt_bf16 under ce_set_attention_waves(16)).
// Requirements beyond attention_vt_launch's: the DMA moves 16-byte pieces - ldk, ldvt multiples of 8 elements and, with batch > 1, a sample
// column stride (= len) that is a multiple of 8; returns CE_ERR_ALIGN otherwise (the caller falls back to the 32 x 32 x 16 kernel).
extern "C" int ce_attn16_launch(const void* Q, const void* K, const void* Vt, int len, int ldk, int ldvt, void* O, int Nq, int H, int ldq, int ldo,
                                float sl2, int batch, int cus, hipStream_t stream) {
  if ((ldk & 7) || (ldvt & 7) || (ldq & 7) || (ldo & 3) || (batch > 1 && (len & 7))) return CE_ERR_ALIGN;
  static bool done_[CE_MAX_DEVICES] = {};
  bool& done = done_[ce_device_slot()];
  if (!done) {
    (void)hipFuncSetAttribute((const void*)attn_fwd_x16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NST * STAGE);
    done = true;
  }
  const int nqb = (Nq + 8 * QW - 1) / (8 * QW);
  const int items = nqb * H * batch;
  // NST * STAGE = 96 KiB of LDS: ONE workgroup fits on a CU, so the persistent grid is one per CU (2 x cus ran as two serial rounds with no
  // tail balancing: ADVICE r5)
  const int grid = items <= cus ? items : (cus & ~7);
  hipLaunchKernelGGL(attn_fwd_x16_kernel, dim3(grid), dim3(512), NST * STAGE, stream, (const bf16*)Q, (const bf16*)K, (const bf16*)Vt, (bf16*)O, Nq, len, H,
                     ldq, ldk, ldvt, ldo, nqb, sl2, batch, len);
  return (int)hipGetLastError();
}
