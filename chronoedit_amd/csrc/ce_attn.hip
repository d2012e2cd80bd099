// Fused (flash-style) attention forward for the DiT blocks, head_dim = 128, bf16, non-causal:
//     O = softmax(Q K^T / sqrt(128)) V                      (K9, self-attention over N tokens)
//     O = bf16(softmax(Q Kt^T) Vt) + bf16(softmax(Q Ki^T) Vi)  (K14, text + image cross-attn)
// Reference: F.scaled_dot_product_attention calls at transformer_chronoedit.py:91-104.
//
// Bound: bf16 MFMA; algorithmic flops = 4 * Nq * Nkv * 128 per head.
//
// Structure (CDNA4 wave64, v_mfma_f32_32x32x16_bf16):
//  * 512-thread workgroup = 8 waves x 32 query rows = 256 query rows of one head; KV tile = 64.
//  * "Swapped" products so the softmax row lives in ONE lane:  S^T = K.Q^T  puts column q = lane&31
//    of the 32x32 accumulator in each lane (16 kv values per 32-kv fragment per lane; the two
//    half-waves hold disjoint kv subsets) -> row max / row sum are in-register + one lane^32 exchange.
//  * O^T = V^T . P^T: the P^T B-operand is taken straight from the S^T accumulator registers
//    (no cross-lane movement): MFMA contracts over "k slots", and we are free to assign slot
//    (half h, j) <-> kv = base + 4h + (j&3) + 8*(j>>2) as long as the V^T A-operand uses the same
//    assignment - it does, via two 8-byte LDS reads per fragment.
//  * K tile row-major in LDS, 16-B chunks XOR-swizzled by row (conflict-free ds_read_b128);
//    V tile transposed in registers (4x4 bf16 blocks) on its way into LDS as V^T[dv][kv] with an
//    8-B-chunk XOR swizzle that is conflict-free for both the ds_write_b64 and the ds_read_b64 side.
//  * K/V tile t+1 is fetched global->VGPR under the MFMAs of tile t and written to the other LDS
//    buffer afterwards (one barrier per tile).
//  * Output rows are staged through LDS and stored as whole 256-B rows (16 B per lane).
//  * blockIdx -> (head, q-block) keeps all q-blocks of a head on one XCD (K/V stay in that L2).
#include <cstdlib>

#include "ce_common.h"

namespace {

constexpr int HD = 128;            // head dim
constexpr int QW = 32;             // query rows per wave
constexpr int KVB = 64;            // kv rows per tile
constexpr int K_TILE_BYTES = KVB * HD * 2;   // 16 KiB, row = 256 B
constexpr int VT_TILE_BYTES = HD * KVB * 2;  // 16 KiB, row (one dv) = 128 B
constexpr int BUF_BYTES = K_TILE_BYTES + VT_TILE_BYTES;
constexpr int OST_ROW = HD * 2 + 16;         // 272-B padded output staging row
// dynamic LDS: the output staging (NWAVE*32 rows x 272 B) overlays the two tile buffers in the 1-segment form and sits
// beside them in the 2-segment form (the segment-0 result waits there while segment 1 streams through the tiles)
constexpr int smem_bytes(int nwave, bool two_seg) {
  const int stage = nwave * QW * OST_ROW;
  return two_seg ? 2 * BUF_BYTES + stage : (stage > 2 * BUF_BYTES ? stage : 2 * BUF_BYTES);
}
constexpr float NEG_BIG = -1.0e30f;

// 8-B chunk swizzle of the V^T rows.  Conflict-free for the ds_write_b64 side (16 contiguous lanes: dv = 4*dvq + j) and the
// ds_read_b64 side (32 lanes: dv = 32*m + lane).  The (dv >> 6) term makes rows 64 apart differ by an XOR instead of a
// constant byte offset, which keeps hipcc from fusing two conflict-free ds_read_b64 into one ds_read2st64_b64 (half
// rate, 32-bank addressing -> 2-way conflicts).
__device__ __forceinline__ int vt_swz(int dv) {
  return ((((dv >> 2) & 7) << 1) | (((dv >> 1) ^ (dv >> 5)) & 1)) ^ (((dv >> 6) & 1) << 1);
}

struct KVSeg {
  const bf16* k;
  const bf16* v;
  int len;
  int ldk;
  int ldv;
};
// Blocked row layout of Q / K / O (V^T form only; rows == 0: plain, sample b's token g in row b N + g): token g of sample b sits in
// row (g / rows) stride + b rows + g % rows - what an all-to-all leaves behind when every rank sent [sample][local token] rows
// (chronoedit_amd/parallel.py: block = source rank, rows = tokens per rank (a multiple of 64), stride = batch * rows).  The V^T
// operand is plain per sample (its producer un-blocks while it transposes): sample b's keys at columns [b vt_cols, ...).
struct BlkRows {
  int rows;        // tokens per block (0: plain layout)
  int stride;      // rows between the starts of consecutive blocks
  uint32_t magic;  // ceil(2^32 / (rows / 64)): tile index -> block by multiply-high (0: rows == 64, block = tile)
  int vt_cols;     // column stride between samples in V^T
  // two-segment V^T form only (ce_attention_2seg_vt_quant_bf16): the output as the out-projection's MX fp8 operand instead of bf16 rows
  unsigned char* q8;  // e4m3 [batch Nq][ld8]
  unsigned char* s8;  // E8M0 block scales, tiled layout of ce_gemm_mxfp8
  int ld8;
};

template <bool TWO_SEG, int NWAVE>
__global__ __launch_bounds__(NWAVE * 64, NWAVE == 8 ? 2 : 2) void attn_fwd_kernel(const bf16* __restrict__ Q, bf16* __restrict__ O, KVSeg seg0,
                                                       KVSeg seg1, int Nq, int H, int ldq, int ldo, int nqb,
                                                       float scale_log2e) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  {  // stacked samples: sample b = blockIdx.y owns rows [b Nq, (b+1) Nq) of Q/O and [b len, (b+1) len) of each K/V segment
    const size_t bz = blockIdx.y;
    Q += bz * Nq * ldq;
    O += bz * Nq * ldo;
    seg0.k += bz * seg0.len * seg0.ldk;
    seg0.v += bz * seg0.len * seg0.ldv;
    if (TWO_SEG) {
      seg1.k += bz * seg1.len * seg1.ldk;
      seg1.v += bz * seg1.len * seg1.ldv;
    }
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
  constexpr int QB = QW * NWAVE;  // query rows per workgroup
  constexpr int NT = NWAVE * 64;  // threads
  const int q0 = qb * QB + wave * QW;
  const int hoff = head * HD;

  // Q^T B-operand fragments: lane (q = l31, half hh) holds Q[q][16*ks + 8*hh .. +8]
  bf16x8 qf[8];
  {
    const bf16* qrow = Q + (size_t)min(q0 + l31, Nq - 1) * ldq + hoff + 8 * hh;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qrow + 16 * ks);
  }

  // staging maps
  // K: 64 rows x 16 chunks(16 B) = 1024 chunks, KREP per thread (rows k_row0 + i * NT/16)
  constexpr int KREP = 1024 / NT, KROWS = NT / 16;
  const int k_ck = tid & 15, k_row0 = tid >> 4;
  // V: 4(kv) x 4(dv) patches, 512 of them, VREP per thread: dv = 4*dvq.., kv = 4*(kvq + i * NT/32)..
  constexpr int VREP = 512 / NT, VKQ = NT / 32;
  const int v_dvq = tid & 31, v_kvq = tid >> 5;

  // output staging rows; in the 2-segment form they live beside the tile buffers so that the
  // segment-0 result can wait there (as bf16) while segment 1 streams through the tiles
  unsigned char* ost = smem + (TWO_SEG ? 2 * BUF_BYTES : 0) + (size_t)(wave * QW + l31) * OST_ROW;

#pragma unroll
  for (int sidx = 0; sidx < (TWO_SEG ? 2 : 1); ++sidx) {
    const KVSeg sg = sidx == 0 ? seg0 : seg1;
    const int ntiles = (sg.len + KVB - 1) / KVB;

    f32x16 oacc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[m][r] = 0.f;
    float m_run = NEG_BIG, l_run = 0.f;

    u32x4 kreg[KREP];
    u32x2 vreg[VREP][4];
    auto load_tile = [&](int t) {
      const int kv0 = t * KVB;
#pragma unroll
      for (int i = 0; i < KREP; ++i) {
        const int r = min(kv0 + k_row0 + KROWS * i, sg.len - 1);
        kreg[i] = *reinterpret_cast<const u32x4*>(sg.k + (size_t)r * sg.ldk + hoff + k_ck * 8);
      }
#pragma unroll
      for (int p = 0; p < VREP; ++p)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = min(kv0 + 4 * (v_kvq + VKQ * p) + i, sg.len - 1);
          vreg[p][i] = *reinterpret_cast<const u32x2*>(sg.v + (size_t)r * sg.ldv + hoff + v_dvq * 4);
        }
    };
    auto store_tile = [&](int buf) {
      unsigned char* sK = smem + buf * BUF_BYTES;
      unsigned char* sV = sK + K_TILE_BYTES;
#pragma unroll
      for (int i = 0; i < KREP; ++i) {
        const int r = k_row0 + KROWS * i;
        *reinterpret_cast<u32x4*>(sK + r * (HD * 2) + ((k_ck ^ (r & 15)) << 4)) = kreg[i];
      }
      // 4x4 transpose of 16-bit elements: vreg[p][i] = {dv0,dv1 | dv2,dv3} of kv row i
#pragma unroll
      for (int p = 0; p < VREP; ++p)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int w = j >> 1;
          uint32_t lo, hi;
          if ((j & 1) == 0) {
            lo = (vreg[p][0][w] & 0xffffu) | (vreg[p][1][w] << 16);
            hi = (vreg[p][2][w] & 0xffffu) | (vreg[p][3][w] << 16);
          } else {
            lo = (vreg[p][0][w] >> 16) | (vreg[p][1][w] & 0xffff0000u);
            hi = (vreg[p][2][w] >> 16) | (vreg[p][3][w] & 0xffff0000u);
          }
          const int dv = 4 * v_dvq + j;
          u32x2 val = {lo, hi};
          *reinterpret_cast<u32x2*>(sV + dv * (KVB * 2) + (((v_kvq + VKQ * p) ^ vt_swz(dv)) << 3)) = val;
        }
    };

    load_tile(0);
    __syncthreads();  // previous segment / nothing: LDS free
    store_tile(0);
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
      const int cur = t & 1;
      if (t + 1 < ntiles) load_tile(t + 1);
      const unsigned char* sK = smem + cur * BUF_BYTES;
      const unsigned char* sV = sK + K_TILE_BYTES;

      // ---- S^T = K . Q^T  (two 32-kv fragments)
      f32x16 st[2];
#pragma unroll
      for (int f = 0; f < 2; ++f) {
#pragma unroll
        for (int r = 0; r < 16; ++r) st[f][r] = 0.f;
        const int r = 32 * f + l31;
        const unsigned char* krow = sK + r * (HD * 2);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(krow + (((2 * ks + hh) ^ (r & 15)) << 4));
          st[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], st[f], 0, 0, 0);
        }
      }
      // ---- mask the kv tail of the last tile: kv = 64 t + 32 f + (r&3) + 8 (r>>2) + 4 hh
      if ((t + 1) * KVB > sg.len) {
        const int base = t * KVB + 4 * hh;
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int kv = base + 32 * f + (r & 3) + 8 * (r >> 2);
            if (kv >= sg.len) st[f][r] = NEG_BIG;
          }
      }
      // ---- online softmax (row = this lane's q; partner lane^32 holds the other kv half)
      float mx = st[0][0];
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[f][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * scale_log2e);
      const float mc = m_new * scale_log2e;
      m_run = m_new;
      float psum = 0.f;
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = __builtin_amdgcn_exp2f(fmaf(st[f][r], scale_log2e, -mc));
          st[f][r] = p;
          psum += p;
        }
      l_run = l_run * alpha + psum;
      if (__any(alpha != 1.0f)) {  // wave-uniform: skip the 64-register O rescale when no row's running max moved
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[m][r] *= alpha;
      }

      // ---- O^T += V^T . P^T : 4 k-steps of 16 kv; step s uses st[s>>1] regs 8*(s&1) .. +8
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int f = s >> 1, rb = 8 * (s & 1);
        f32x2 t0 = {st[f][rb + 0], st[f][rb + 1]}, t1 = {st[f][rb + 2], st[f][rb + 3]};
        f32x2 t2 = {st[f][rb + 4], st[f][rb + 5]}, t3 = {st[f][rb + 6], st[f][rb + 7]};
        const bf16x2 p0 = __builtin_convertvector(t0, bf16x2), p1 = __builtin_convertvector(t1, bf16x2);
        const bf16x2 p2 = __builtin_convertvector(t2, bf16x2), p3 = __builtin_convertvector(t3, bf16x2);
        const bf16x8 pf = {p0[0], p0[1], p1[0], p1[1], p2[0], p2[1], p3[0], p3[1]};
        const int c0 = 4 * s + hh;  // 8-B chunk index of kv = 16 s + 4 hh
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int dv = 32 * m + l31;
          const unsigned char* vrow = sV + dv * (KVB * 2);
          const int sw = vt_swz(dv);
          const bf16x4 va = *reinterpret_cast<const bf16x4*>(vrow + ((c0 ^ sw) << 3));
          const bf16x4 vb = *reinterpret_cast<const bf16x4*>(vrow + (((c0 + 2) ^ sw) << 3));
          const bf16x8 vf = {va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
          oacc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oacc[m], 0, 0, 0);
        }
      }
      if (t + 1 < ntiles) store_tile(cur ^ 1);
      __syncthreads();
    }

    // ---- finalize this segment: normalise, round to bf16 (SDPA output dtype)
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    // stage this wave's 32 x 128 tile: lane (q = l31, half hh) owns dv = 32 m + 8 a + 4 hh + b
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        uint32_t w0 = pack_bf16(oacc[m][4 * a + 0] * inv, oacc[m][4 * a + 1] * inv);
        uint32_t w1 = pack_bf16(oacc[m][4 * a + 2] * inv, oacc[m][4 * a + 3] * inv);
        u32x2* slot = reinterpret_cast<u32x2*>(ost + (32 * m + 8 * a + 4 * hh) * 2);
        if (TWO_SEG && sidx == 1) {  // bf16 add of the two SDPA outputs (transformer_chronoedit.py:104)
          const u32x2 pv = *slot;    // this lane's own segment-0 value
          w0 = pack_bf16(bf16lo(pv[0]) + bf16lo(w0), bf16hi(pv[0]) + bf16hi(w0));
          w1 = pack_bf16(bf16lo(pv[1]) + bf16lo(w1), bf16hi(pv[1]) + bf16hi(w1));
        }
        u32x2 val = {w0, w1};
        *slot = val;
      }
  }

  __syncthreads();
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = lane + 64 * i;
    const int rl = c >> 4, cc = c & 15;
    const int q = q0 + rl;
    if (q < Nq) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(smem + (TWO_SEG ? 2 * BUF_BYTES : 0) + (size_t)(wave * QW + rl) * OST_ROW + cc * 16);
      *reinterpret_cast<u32x4*>(O + (size_t)q * ldo + hoff + cc * 8) = v;
    }
  }
}


// ------------------------------------------------------------------------------------------------
// LDS images of the software-pipelined kernel: PADDED rows instead of XOR swizzles, so that every fragment read of a tile is
// "one per-lane base + an immediate offset" (the XOR form cost ~150 address VALU ops per tile and wave - more VALU issue
// time than the softmax itself; PMC: VALU busy 49 % vs MFMA busy 39 %).
//   K   [64 kv][272 B]: ds_read_b128 of 16 rows (distinct mod 16) at one chunk hit slots (r + c) mod 16 - conflict-free.
//   V^T [128 dv][144 B]: within every 16-kv k-step the 4-kv chunks are stored in the order (h, b) instead of (b, h)
//   (kv = 16 s + 8 b + 4 h + i), so the 8 values of an MFMA A-fragment (b = 0,1 for this lane's h) are 16 contiguous
//   bytes: ONE ds_read_b128 per fragment (hipcc fused the two ds_read_b64 of the plain layout into half-rate
//   ds_read2_b64).  16 lanes of a b128 group (dv distinct mod 16... x9) hit 16 distinct slots: conflict-free; the
//   ds_write_b64 side is conflict-free through the thread -> patch map (below).
// ------------------------------------------------------------------------------------------------
#define CE_EPOCH_BARRIER()                          \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
  __builtin_amdgcn_s_barrier()
constexpr int PK_ROW = HD * 2 + 16;           // 272
constexpr int PV_ROW = KVB * 2 + 16;          // 144
constexpr int PK_TILE = KVB * PK_ROW;         // 17408
constexpr int PV_TILE = HD * PV_ROW;          // 18432
typedef __attribute__((ext_vector_type(4))) unsigned pp_u4;
typedef __attribute__((ext_vector_type(2))) unsigned pp_u2;

// ------------------------------------------------------------------------------------------------
// Software-pipelined variant on the padded LDS images ("sp").  Measured fact it is built on (tools/probes/pipe_probe.hip):
// on one SIMD, one wave's MFMAs and ANOTHER wave's VALU work serialise at equal priority (16 MFMA + 32 v_exp per
// iteration: 5.3 ms + 3.2 ms alone, 8.7 ms together), while the same MFMAs and v_exp interleaved in ONE wave's
// instruction stream cost the MFMA time alone (5.6 ms; two such waves per SIMD: 10.1 ms = the matrix pipe saturated).
// Round 1's ping-pong kernel (two wave groups one barrier apart; removed) relied on the cross-wave overlap and its s_memtime
// stamps showed epochs of ~1900 cycles for 1024 cycles of MFMA per SIMD.  Here every wave runs the same program and hides its
// own VALU:
//     iteration t:  vmcnt(4) (this wave's pieces of K(t) have landed), barrier
//                   | S(t) = K(t).Q^T: 16 MFMA, alternating accumulators, one scheduling region each; in their gaps the
//                     LDS-DMA of K(t+1), V(t+1) registers -> LDS (transposed 4x4 patches), the fetch of V(t+2)
//                   | tail mask (last tile), exact offset (first tile only), O *= alpha(t-1) (rare)
//                   | O += V^T(t-1).P^T(t-1): 16 MFMA, each followed by 2 v_exp + 2 adds of P(t) and one packed bf16
//                     conversion that replaces a word of P(t-1) in place
//                   | row-sum check of the speculative softmax; exact route (S(t) again, true max, offset moved) if it fails
// LDS: K double-buffered (LDS-DMA image, swizzled on the source side), V^T triple-buffered (V(t-1) is read while V(t+1) is
// written): one barrier per tile.  History and A/B numbers of each step: DESIGN.md section 4.2.
// ------------------------------------------------------------------------------------------------
constexpr int K_GRP = 1088;  // LDS-DMA K image: 4 rows (1 KiB) + 64 B pad per group; 16 groups = PK_TILE
typedef __attribute__((address_space(3))) void lds_void;
#ifdef CE_DIAGNOSTICS
__device__ unsigned long long g_sp_exact_hits = 0ull;
#endif
constexpr float SP_SPEC_THR = 1024.0f;  // a lane's partial row sum above this sends the tile through the exact route
constexpr int SP_V0 = 2 * PK_TILE;                       // V^T buffers follow the two K buffers
constexpr int SP_TILE_BYTES = 2 * PK_TILE + 3 * PV_TILE;  // 90112 (also holds the 69632-B O staging)
constexpr int sp_smem_bytes(bool two_seg) { return two_seg ? SP_TILE_BYTES + 8 * QW * OST_ROW : SP_TILE_BYTES; }

template <bool TWO_SEG, bool VT = false, bool QOUT = false>  // QOUT: the output as an MX fp8 operand (two-segment V^T form only)
__global__ __launch_bounds__(512, 2) void attn_fwd_sp_kernel(const bf16* __restrict__ Q_, bf16* __restrict__ O_, KVSeg seg0_,
                                                              KVSeg seg1_, int Nq, int H, int ldq, int ldo, int nqb,
                                                              float scale_log2e, int batch, BlkRows blk) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NWAVE = 8, QB = QW * NWAVE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  // Persistent form: a workgroup walks the work order with stride gridDim.x (a multiple of 8: the head stays on its XCD).  Launched
  // with one workgroup per item the loop runs once; the V^T launcher uses 2 x #CUs workgroups (A/B, profiles/r02_attention_vt_knobs_ab.txt:
  // +1.5 ... 3 % over one workgroup per item; #CUs, 3 x and 4 x #CUs are level or worse, 1.5 x #CUs -14 %).
  for (int item = blockIdx.x; item < nqb * H * batch; item += gridDim.x) {
  const bf16* Q = Q_;
  bf16* O = O_;
  KVSeg seg0 = seg0_, seg1 = seg1_;

  // Work order (batch folded into item): every XCD takes its heads' FULL 256-row query blocks first, sample by sample,
  // and the remainder blocks (Nq % 256 rows; only their first waves have work and the others merely stage, so they run
  // ~2.5x faster) last - they fill the partially occupied final round of workgroups instead of heading it.  At
  // Nq = 7200, H = 40, two samples: 2320 workgroups = 9.06 rounds of 256 CUs used to cost ten.
  int head, qb, bz;
  {
    const int nqb_full = Nq / QB;
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
  {  // stacked samples: sample bz owns rows [bz Nq, (bz+1) Nq) of Q/O and [bz len, (bz+1) len) of each K/V segment
    if (VT && blk.rows > 0) {  // blocked rows: sample bz starts bz * rows into every block
      Q += (size_t)bz * blk.rows * ldq;
      O += (size_t)bz * blk.rows * ldo;
      seg0.k += (size_t)bz * blk.rows * seg0.ldk;
      seg0.v += (size_t)bz * blk.vt_cols;
    } else {
      Q += (size_t)bz * Nq * ldq;
      O += (size_t)bz * Nq * ldo;
      seg0.k += (size_t)bz * seg0.len * seg0.ldk;
      // VT: the samples' keys sit side by side in the columns of V^T (two-segment form: at the column strides blk.vt_cols / blk.stride)
      seg0.v += VT ? (size_t)bz * (TWO_SEG ? blk.vt_cols : seg0.len) : (size_t)bz * seg0.len * seg0.ldv;
    }
    if (TWO_SEG) {
      seg1.k += (size_t)bz * seg1.len * seg1.ldk;
      seg1.v += VT ? (size_t)bz * blk.stride : (size_t)bz * seg1.len * seg1.ldv;
    }
  }
  const int q0 = qb * QB + wave * QW;
  const int hoff = head * HD;
  const bool active = q0 < Nq;  // wave-uniform: a wave past the last query row only helps staging the K / V tiles
  // row of this wave's first query token (blocked layout: a wave's 32 tokens never straddle a block - blocks are multiples of 64 rows)
  const int q0row = (VT && blk.rows > 0) ? (q0 / blk.rows) * blk.stride + q0 % blk.rows : q0;

  bf16x8 qf[8];
  {
    const bf16* qrow = Q + (size_t)(q0 + l31 < Nq ? q0row + l31 : 0) * ldq + hoff + 8 * hh;  // (rows past Nq: any valid row, never stored)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qrow + 16 * ks);
  }
  auto scale_q = [&]() __attribute__((always_inline)) {
  // Q is scaled ONCE by softmax_scale * log2(e) (fp32 multiply, rounded back to bf16): S comes out of the matrix pipe in
  // the exp2 domain and, with the running max folded into the accumulator init below, P is a bare v_exp_f32 per element.
  // (Plain VALU ops beside MFMAs are not hidden on this chip - probe in tools/probes - and the fma per element was a
  // quarter of them.  torch's math SDPA also scales q and k in bf16 before the product.)
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    const u32x4 w = __builtin_bit_cast(u32x4, qf[ks]);
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = pack_bf16(bf16lo(w[i]) * scale_log2e, bf16hi(w[i]) * scale_log2e);
    qf[ks] = __builtin_bit_cast(bf16x8, o);
  }
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) asm volatile("" : "+v"(qf[ks]));  // retire the Q loads before the loop (else hipcc keeps them pending into it)

  };
  // staging shares and LDS bases: V^T rows of 144 B in (h, b) chunk order (above), K by LDS-DMA (below)
  const int v_dvq = (tid & 1) | (((tid >> 4) & 15) << 1), v_kvq = ((tid >> 1) & 7) | (((tid >> 8) & 1) << 3);
  const int v_chunk = (v_kvq & ~3) | ((v_kvq & 1) << 1) | ((v_kvq >> 1) & 1);
  unsigned char* ost = smem + (TWO_SEG ? SP_TILE_BYTES : 0) + (size_t)(wave * QW + l31) * OST_ROW;
  // K image filled by LDS-DMA (no VGPR round trip, no ds_write): 16 groups of 4 kv rows, one 1 KiB wave-instruction each,
  // 1088 B apart; in a group, row w = r & 3 keeps its 16-B chunk c in slot c ^ w.  A b128 read group covers 4 groups
  // (x 64 B pad = slots +0, +4, +8, +12) times 4 rows (low slot bits ^ w): 16 distinct slots, conflict-free.  With
  // c = 2 ks + h the slot is 4 (ks >> 1) + 2 ((ks & 1) ^ (w >> 1)) + (h ^ (w & 1)): two per-lane bases (even / odd ks) +
  // immediates.
  const int k_w = l31 & 3;
  const unsigned char* k_rd = smem + (l31 >> 2) * K_GRP + k_w * 256 + ((hh ^ (k_w & 1)) << 4);  // + buf*PK_TILE + f*8*K_GRP
  const int k_eo[2] = {(k_w >> 1) << 5, ((k_w >> 1) ^ 1) << 5};                                    // + k_eo[ks & 1] + (ks >> 1)*64
  const unsigned char* v_rd = smem + SP_V0 + l31 * PV_ROW + hh * 16;    // + buf*PV_TILE + m*32*PV_ROW + s*32
  // VT (V arrives transposed, [head_dim row][keys], and its tiles come by LDS-DMA like K's): image = 128 rows x 128 B, lane-linear
  // per 1 KiB wave-instruction (8 rows), chunk c of row d kept in slot c ^ ((d >> 1) & 7) on the SOURCE side: the 16 lanes of a
  // ds_read_b128 group (16 consecutive d, one chunk) hit slots 8 (d & 1) + (c ^ (d >> 1 & 7)) - all distinct.  With c = 2 s + h the
  // slot is 2 (s ^ x >> 1) + (h ^ x & 1), x = (d >> 1) & 7: four per-lane bases (one per k-step) + immediates.
  int vt_off[4];
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) {
    const int x = (l31 >> 1) & 7;
    vt_off[s4] = l31 * 128 + ((2 * (s4 ^ (x >> 1)) + (hh ^ (x & 1))) << 4);
  }
  unsigned char* v_wr = smem + SP_V0 + (4 * v_dvq) * PV_ROW + v_chunk * 8;  // + buf*PV_TILE + j*PV_ROW

#pragma unroll
  for (int sidx = 0; sidx < (TWO_SEG ? 2 : 1); ++sidx) {
    const KVSeg sg = sidx == 0 ? seg0 : seg1;
    const int ntiles = (sg.len + KVB - 1) / KVB;
    const int k_rows_span = (VT && blk.rows > 0) ? ((sg.len - 1) / blk.rows) * blk.stride + blk.rows : sg.len;  // rows from the first to the last key's
    const auto k_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(sg.k + hoff), 0, (k_rows_span - 1) * sg.ldk * 2 + HD * 2, 0x00020000);
    // VT: rows hoff .. hoff + 127 of V^T [H * 128][ldv]; a tile is a 128-B column strip.  Columns past this sample's keys hold the
    // next sample's keys or the (finite, zeroed) padding of the buffer: they only ever meet P = 0.
    const auto v_rsrc = VT ? __builtin_amdgcn_make_buffer_rsrc((void*)(sg.v + (size_t)hoff * sg.ldv), 0, ((HD - 1) * sg.ldv + ntiles * KVB) * 2, 0x00020000)
                           : __builtin_amdgcn_make_buffer_rsrc((void*)(sg.v + hoff), 0, (sg.len - 1) * sg.ldv * 2 + HD * 2, 0x00020000);
    int v_voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v_voff[i] = (4 * v_kvq + i) * sg.ldv * 2 + v_dvq * 8;
    const int k_tile_bytes = KVB * sg.ldk * 2, v_tile_bytes = KVB * sg.ldv * 2;

    f32x16 oacc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[m][r] = 0.f;
    // mc: offset in use (exp2 domain); the K.Q^T accumulators start at -mc (cinit), so S arrives as "score - mc"
    float l_run = 0.f, alpha_prev = 1.0f, mc = 0.f;
    f32x16 cinit;
#pragma unroll
    for (int r = 0; r < 16; ++r) cinit[r] = 0.f;
    u32x4 ppk[4];  // P(t-1) as bf16 pairs: word w of ppk[s] = elements 8 s + 2 w, + 1 of the lane's 32 scores
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) ppk[s4] = u32x4{0u, 0u, 0u, 0u};

    pp_u2 vreg[4];
    // wave-instruction j fills group wave + 8 j of the image: lane -> row 4 (wave + 8 j) + (lane >> 4), chunk (lane & 15) ^ (lane >> 4)
    // VT: image row rho holds key pi(rho) = rho with bits 2 and 3 swapped, so that the 8 keys a lane half owns per k-step of P.V
    // (accumulator rows 16 s + 8 b + 4 h + i) are the CONTIGUOUS keys 16 s + 8 h + 4 b + i: its V^T fragment is one 16-B chunk of
    // the natural layout (the register-staged path gets the same effect by permuting V while it transposes it).
    const int kd_row = VT ? (lane >> 4) + 4 * ((wave >> 1) & 1) + 8 * (wave & 1) + 16 * (wave >> 2) : 4 * wave + (lane >> 4);
    const int kd_voff0 = kd_row * sg.ldk * 2 + (((lane & 15) ^ (lane >> 4)) << 4), kd_voff1 = kd_voff0 + 32 * sg.ldk * 2;
    // V^T piece j of this wave: rows 8 (wave + 8 j) + (lane >> 3), slot lane & 7 <- chunk slot ^ ((row >> 1) & 7)
    const int vd_row = 8 * wave + (lane >> 3);
    const int vd_voff0 = vd_row * sg.ldv * 2 + (((lane & 7) ^ ((vd_row >> 1) & 7)) << 4), vd_voff1 = vd_voff0 + 64 * sg.ldv * 2;
    auto dma_v = [&](int t, int buf, int j) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(v_rsrc, (lds_void*)(smem + SP_V0 + buf * PV_TILE + (wave + 8 * j) * 1024), 16,
                                               j ? vd_voff1 : vd_voff0, t * (KVB * 2), 0, 0);
    };
    auto dma_k = [&](int t, int buf, int j) {
      int soff = t * k_tile_bytes;
      if (VT && blk.rows > 0) {  // key tile t lives in block t / (rows / 64): scalar multiply-high, no per-lane work
        const int b_ = blk.magic ? (int)__umulhi((uint32_t)t, blk.magic) : t;  // (magic == 0: one tile per block)
        soff = (b_ * blk.stride + (t - b_ * (blk.rows >> 6)) * KVB) * sg.ldk * 2;
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(k_rsrc, (lds_void*)(smem + buf * PK_TILE + (wave + 8 * j) * K_GRP), 16,
                                               j ? kd_voff1 : kd_voff0, soff, 0, 0);
    };
    auto load_v = [&](int t) {
      const int so = t * v_tile_bytes;
#pragma unroll
      for (int i = 0; i < 4; ++i) vreg[i] = __builtin_amdgcn_raw_buffer_load_b64(v_rsrc, v_voff[i], so, 0);
    };
    auto store_v_part = [&](int buf, int j) {  // row j of this thread's transposed 4x4 patch
      const int w = j >> 1;
      uint32_t lo, hi;
      if ((j & 1) == 0) {
        lo = (vreg[0][w] & 0xffffu) | (vreg[1][w] << 16);
        hi = (vreg[2][w] & 0xffffu) | (vreg[3][w] << 16);
      } else {
        lo = (vreg[0][w] >> 16) | (vreg[1][w] & 0xffff0000u);
        hi = (vreg[2][w] >> 16) | (vreg[3][w] & 0xffff0000u);
      }
      pp_u2 val = {lo, hi};
      *reinterpret_cast<pp_u2*>(v_wr + buf * PV_TILE + j * PV_ROW) = val;
    };
    auto store_v = [&](int buf) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int w = j >> 1;
        uint32_t lo, hi;
        if ((j & 1) == 0) {
          lo = (vreg[0][w] & 0xffffu) | (vreg[1][w] << 16);
          hi = (vreg[2][w] & 0xffffu) | (vreg[3][w] << 16);
        } else {
          lo = (vreg[0][w] >> 16) | (vreg[1][w] & 0xffff0000u);
          hi = (vreg[2][w] >> 16) | (vreg[3][w] & 0xffff0000u);
        }
        pp_u2 val = {lo, hi};
        *reinterpret_cast<pp_u2*>(v_wr + buf * PV_TILE + j * PV_ROW) = val;
      }
    };

    // ---- prologue: tile 0 -> K buffer 0 / V buffer 0; V buffer 2 plays "V(-1)" (zeros: P(-1) = 0 must not meet NaNs)
    if (TWO_SEG && sidx == 1) { CE_EPOCH_BARRIER(); }  // segment 0's drain still read the V buffers
    dma_k(0, 0, 0);
    dma_k(0, 0, 1);
    __builtin_amdgcn_sched_barrier(0);
    if (VT) {
      dma_v(0, 0, 0);
      dma_v(0, 0, 1);
    } else {
      load_v(0);
    }
    {
      const pp_u4 z = {0u, 0u, 0u, 0u};
      for (int i = tid; i < PV_TILE / 16; i += NWAVE * 64) *reinterpret_cast<pp_u4*>(smem + SP_V0 + 2 * PV_TILE + i * 16) = z;
    }
    if (!VT) {
      store_v(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // K(0) has landed (its DMA precedes the V(0) fetch just consumed)
      load_v(1);
    }
    // the Q rows were requested before the first tile DMAs: waiting for them here (not before issuing the DMAs) overlaps the two
    // latencies of an item's prologue (+0.3 ... 0.5 % in the A/B of profiles/r02_attention_vt_knobs_ab.txt)
    if (sidx == 0) scale_q();

    // From here on the vector-memory queue of a wave holds, in order: K(t) DMA x2 (tile t-1, units 1 and 3), V(t+1) x4
    // (tile t-1, unit 12): "vmcnt(4)" in front of the barrier of tile t = K(t) is in LDS.
    // VT: the queue holds K(t) DMA x2 (units 1, 3 of tile t-1), V(t) DMA x2 (units 5, 7): "vmcnt(2)" = K(t) and V(t-1) are in LDS;
    // V(t) is first read one iteration later (P(t).V(t) runs inside tile t+1) and may still be in flight.
#define CE_SP_KWAIT()                                          \
  do {                                                         \
    if (VT) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");   \
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      \
  } while (0)
    int vb_prev = 2, vb_cur = 0;  // V^T buffer of tile t-1 / tile t

    // A wave without query rows only stages, behind the same barriers - in a loop of its own: as a `continue` path inside
    // the main loop its fetches joined the main path's in phi nodes that hipcc resolved with six v_mov_b64 (behind a
    // vmcnt(0)) in the loop latch of every wave.
    for (int t = 0; !active && t < ntiles; ++t) {
      CE_SP_KWAIT();
      CE_EPOCH_BARRIER();
      const int vb_next = 3 - vb_prev - vb_cur;
      dma_k(t + 1, (t + 1) & 1, 0);
      dma_k(t + 1, (t + 1) & 1, 1);
      __builtin_amdgcn_sched_barrier(0);
      if (VT) {
        dma_v(t + 1, vb_next, 0);
        dma_v(t + 1, vb_next, 1);
      } else {
        store_v(vb_next);
        load_v(t + 2);
      }
      vb_prev = vb_cur;
      vb_cur = vb_next;
    }
    for (int t = 0; active && t < ntiles; ++t) {
      CE_SP_KWAIT();
      CE_EPOCH_BARRIER();  // K(t), V(t) visible; K(t-1) and V(t-2) no longer read by anyone
      const int vb_next = 3 - vb_prev - vb_cur;
      const unsigned char* kb = k_rd + (t & 1) * PK_TILE;
      const unsigned char* vb = v_rd + vb_prev * PV_TILE;
      const unsigned char* vtb = smem + SP_V0 + vb_prev * PV_TILE;  // VT image of tile t-1

      // ---- S^T(t) = K(t).Q^T: MFMA i works on kv fragment f = i & 1, k-step ks = i >> 1 (alternating accumulators; every
      // unit - MFMA, ring refill, its piece of staging - is its own scheduling region: with sched_group_barrier alone hipcc
      // re-linearised the MFMAs accumulator by accumulator, and a filler between two MFMAs on the SAME accumulator costs
      // ~40 cycles (MI355X guide, latency table) - pinning the alternating order was worth 4 % of the kernel).
      // The staging of the NEXT tiles rides inside this MFMA stream instead of in front of it (stamps: with all eight
      // waves storing right after the barrier the matrix pipe idled ~600 cycles per tile): K(t+1) / V(t+1) registers
      // -> LDS after MFMAs 1, 3, 5..8, fetch of K(t+2) / V(t+2) after MFMAs 10 and 12.  Unconditional: past the last
      // tile the stores fill buffers nobody reads and the fetches are out of range of the buffer descriptor (zeros).
      const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      f32x16 st[2];
      {
        constexpr int RING = VT ? 4 : 6;  // K fragments in flight (A/B of 2 / 3 / 4 / 6 / 8 with the DMA-staged V: all within 0.8 %, 4 best)
        bf16x8 kf[RING];
#define CE_LDK(i) (*reinterpret_cast<const bf16x8*>(kb + k_eo[((i) >> 1) & 1] + ((i) & 1) * 8 * K_GRP + ((i) >> 2) * 64))
#pragma unroll
        for (int i = 0; i < RING; ++i) kf[i] = CE_LDK(i);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          if (i < 2) {
            // untied form spelled out (D != C, both chains start from the same cinit registers): left to itself hipcc
            // tied D to C for one of the two and copied cinit first, 8 v_mov_b64 per tile
            asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(st[i & 1]) : "v"(kf[i % RING]), "v"(qf[i >> 1]), "v"(cinit));
          } else {
            st[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[i % RING], qf[i >> 1], st[i & 1], 0, 0, 0);
          }
          if (i + RING < 16) kf[i % RING] = CE_LDK(i + RING);
          if (i == 1) dma_k(t + 1, (t + 1) & 1, 0);
          if (i == 3) dma_k(t + 1, (t + 1) & 1, 1);
          if (VT) {
            if (i == 5) dma_v(t + 1, vb_next, 0);
            if (i == 7) dma_v(t + 1, vb_next, 1);
          } else {
            if (i == 12) load_v(t + 2);
            if (i >= 5 && i < 9) store_v_part(vb_next, i - 5);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
#undef CE_LDK
      }
      // ---- softmax of tile t, SPECULATIVE on the offset mc in use: P(t) = exp2(S(t)) is formed straight away (S arrives as
      // "score - mc" from the accumulator init) beside the P(t-1).V(t-1) MFMAs, with no row max in front of it - the 23
      // v_max of a tile were a quarter of the VALU work that nothing hides on this chip (probe in tools/probes).  Only the
      // row sums, which are needed anyway, are checked afterwards: a partial sum above 2^10 (or inf) in any lane means some
      // row's scores climbed more than ~10 octaves over its offset; that wave then recomputes S(t) from the K tile still
      // in LDS, moves the offset to the true max and redoes the tile's softmax (exact_tile below).  Nothing of a failed
      // attempt survives: P(t) meets V(t) only in the next iteration and l is updated after the check.  The first tile
      // always takes the exact route (offset = its row max, P <= 1); afterwards P <= 2^10, harmless in fp32 / bf16.
      float alpha = 1.0f;
      float psum = 0.f;
      auto mask_tail = [&]() {
        if ((t + 1) * KVB > sg.len) {
          // keys of this tile past the segment's end: element (f, r) of the lane holds key t KVB + c(f, r) + (VT ? 8 : 4) hh, masked when
          // c >= thr.  thr passes through an opaque asm INSIDE the branch: without it hipcc if-converted the whole block - 32 v_cmp +
          // 31 v_cndmask issued in every tile of every wave (a third of the loop's plain VALU work), for a mask that applies to the
          // last tile only (round 4; found in the ISA of the one-wave-per-SIMD kernel, where the compares were hoisted and spilled).
          int thr = sg.len - t * KVB - (VT ? 8 : 4) * hh;
          asm volatile("" : "+v"(thr));
#pragma unroll
          for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int c = VT ? 32 * f + 16 * (r >> 3) + 4 * ((r >> 2) & 1) + (r & 3)  // pi(row), see dma_k
                               : 32 * f + (r & 3) + 8 * (r >> 2);
              if (c >= thr) st[f][r] = NEG_BIG;
            }
        }
      };
      auto rebase = [&]() {  // exact tile max of both halves of the row (lane ^ 32 holds the other half): move the offset
        float mx;
        {
          float m0 = fmaxf(st[0][0], st[0][1]), m1 = fmaxf(st[0][8], st[0][9]), m2 = fmaxf(st[1][0], st[1][1]), m3 = fmaxf(st[1][8], st[1][9]);
#pragma unroll
          for (int r = 2; r < 8; ++r) {
            m0 = fmaxf(m0, st[0][r]);
            m1 = fmaxf(m1, st[0][8 + r]);
            m2 = fmaxf(m2, st[1][r]);
            m3 = fmaxf(m3, st[1][8 + r]);
          }
          mx = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
        }
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
        mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        const float shift = t == 0 ? mx : fmaxf(mx, 0.f);  // the offset only ever rises after the first tile
        alpha = __builtin_amdgcn_exp2f(-shift);
        mc += shift;
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
          for (int r = 0; r < 16; ++r) st[f][r] -= shift;
#pragma unroll
        for (int r = 0; r < 16; ++r) cinit[r] = -mc;
      };
      auto pack_pair = [&](int j) {  // elements 2j, 2j+1 of P(t) -> word j & 3 of the P fragment of k-step j >> 2
        const f32x2 pr = {st[j >> 3][(2 * j) & 15], st[j >> 3][(2 * j + 1) & 15]};
        ppk[j >> 2][j & 3] = __builtin_bit_cast(uint32_t, __builtin_convertvector(pr, bf16x2));
      };
      mask_tail();
      if (t == 0) rebase();
      // O at the scale of m(t-1) before P(t-1).V(t-1) is added (rare after the first tiles)
      if (__any(alpha_prev != 1.0f)) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[m][r] *= alpha_prev;
      }
      // ---- O^T += V^T(t-1).P^T(t-1) on the matrix pipe, P(t) on the VALU, one stream: unit u = (k-step u >> 2, dv
      // fragment u & 3) is one MFMA, the v_exp of elements 2u, 2u+1, two adds into the row sum (one unit behind, for
      // the transcendental result latency) and one packed bf16 conversion (four units behind: pair j lands in the word of
      // the P fragment that MFMA j + 3 was the last to read, so P(t) replaces P(t-1) in place, without copies).
      {
#define CE_LDV(u)                                                                                         \
  (VT ? *reinterpret_cast<const bf16x8*>(vtb + vt_off[(u) >> 2] + ((u) & 3) * 4096)                       \
      : *reinterpret_cast<const bf16x8*>(vb + ((u) & 3) * 32 * PV_ROW + ((u) >> 2) * 32))
        constexpr int VRING = 6;
        bf16x8 vf[VRING];
#pragma unroll
        for (int u = 0; u < VRING; ++u) vf[u] = CE_LDV(u);
        // One scheduling region per unit (sched_barrier(0) after each): hipcc may order the instructions of a unit as it
        // likes but cannot pull the exp2 work of several units together (with sched_group_barrier alone it issued six
        // MFMAs back to back and then bursts of eight or nine v_exp, longer than an MFMA's 32-cycle shadow).
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          oacc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[u % VRING], __builtin_bit_cast(bf16x8, ppk[u >> 2]), oacc[u & 3], 0, 0, 0);
          if (u + VRING < 16) vf[u % VRING] = CE_LDV(u + VRING);
#pragma unroll
          for (int e = 2 * u; e < 2 * u + 2; ++e)
            st[e >> 4][e & 15] = __builtin_amdgcn_exp2f(st[e >> 4][e & 15]);
          if (u > 0) {  // (scalar adds: v_pk_add_f32 beside MFMAs is slower than the two adds it replaces; measured +1.6 %)
            psum += st[(u - 1) >> 3][(2 * u - 2) & 15];
            psum += st[(u - 1) >> 3][(2 * u - 1) & 15];
          }
          if (u >= 4) pack_pair(u - 4);
          __builtin_amdgcn_sched_barrier(0);
        }
        psum += st[1][14];
        psum += st[1][15];
#pragma unroll
        for (int j = 12; j < 16; ++j) pack_pair(j);
#undef CE_LDV
      }
      if (__builtin_expect(t > 0 && __any(psum > SP_SPEC_THR), 0)) {
        // exact_tile: S(t) again (K(t) is untouched until the next barrier), true row max, plain softmax
        CE_DIAG_COUNT_EXACT(g_sp_exact_hits);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const bf16x8 kfr = *reinterpret_cast<const bf16x8*>(kb + k_eo[(i >> 1) & 1] + (i & 1) * 8 * K_GRP + (i >> 2) * 64);
          st[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr, qf[i >> 1], i < 2 ? zero16 : st[i & 1], 0, 0, 0);
        }
#pragma unroll
        for (int f = 0; f < 2; ++f)  // (zero init and an explicit "- mc": a second live copy of cinit cost the common path 8 v_mov)
#pragma unroll
          for (int r = 0; r < 16; ++r) st[f][r] -= mc;
        mask_tail();
        rebase();
        psum = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          st[j >> 3][(2 * j) & 15] = __builtin_amdgcn_exp2f(st[j >> 3][(2 * j) & 15]);
          st[j >> 3][(2 * j + 1) & 15] = __builtin_amdgcn_exp2f(st[j >> 3][(2 * j + 1) & 15]);
          psum += st[j >> 3][(2 * j) & 15] + st[j >> 3][(2 * j + 1) & 15];
          pack_pair(j);
        }
      }
      l_run = l_run * alpha + psum;
      alpha_prev = alpha;
      vb_prev = vb_cur;
      vb_cur = vb_next;
    }
    // ---- drain: P(ntiles-1).V(ntiles-1); its V buffer (now vb_prev) became visible at the last barrier
    if (active && __any(alpha_prev != 1.0f)) {
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[m][r] *= alpha_prev;
    }
    if (VT) {  // V(ntiles-1) was issued during the last-but-one tile and may still be in flight: land it, for everybody
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    if (active) {
      const unsigned char* vb = v_rd + vb_prev * PV_TILE;
      const unsigned char* vtb = smem + SP_V0 + vb_prev * PV_TILE;
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const bf16x8 vfd = VT ? *reinterpret_cast<const bf16x8*>(vtb + vt_off[u >> 2] + (u & 3) * 4096)
                              : *reinterpret_cast<const bf16x8*>(vb + (u & 3) * 32 * PV_ROW + (u >> 2) * 32);
        oacc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfd, __builtin_bit_cast(bf16x8, ppk[u >> 2]), oacc[u & 3], 0, 0, 0);
      }
    }
    CE_EPOCH_BARRIER();  // every wave is done with the tile buffers (the O staging overlays them when !TWO_SEG)

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        uint32_t w0 = pack_bf16(oacc[m][4 * a + 0] * inv, oacc[m][4 * a + 1] * inv);
        uint32_t w1 = pack_bf16(oacc[m][4 * a + 2] * inv, oacc[m][4 * a + 3] * inv);
        u32x2* slot = reinterpret_cast<u32x2*>(ost + (32 * m + 8 * a + 4 * hh) * 2);
        if (TWO_SEG && sidx == 1) {
          const u32x2 pv = *slot;
          w0 = pack_bf16(bf16lo(pv[0]) + bf16lo(w0), bf16hi(pv[0]) + bf16hi(w0));
          w1 = pack_bf16(bf16lo(pv[1]) + bf16lo(w1), bf16hi(pv[1]) + bf16hi(w1));
        }
        u32x2 val = {w0, w1};
        *slot = val;
      }
  }

  if (TWO_SEG && VT && QOUT) {
    // MX fp8 output (fp8 mode's cross-attention feeding ce_gemm_mxfp8): the bf16 rows just staged, quantised as ce_quant_rows_mxfp8 would -
    // a 32-channel block m of a query row is this lane's 16 values and the 16 of lane ^ 32.  The e4m3 bytes of block m overwrite bytes
    // [32 m, 32 m + 32) of the row's staging area: only bf16 slots of blocks <= m, which this lane pair has read by then.
    const int ktiles = (H * HD) >> 7;
    const int grow = bz * Nq + q0 + l31;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      uint32_t pk[8];
      float amax = 0.f;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const u32x2 pv = *reinterpret_cast<const u32x2*>(ost + (32 * m + 8 * a + 4 * hh) * 2);
        pk[2 * a] = pv[0];
        pk[2 * a + 1] = pv[1];
        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(bf16lo(pv[0])), fabsf(bf16hi(pv[0]))), fmaxf(fabsf(bf16lo(pv[1])), fabsf(bf16hi(pv[1])))));
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
      if (hh == 0 && q0 + l31 < Nq) blk.s8[mx_gemm_scale_offset(grow, head * 4 + m, ktiles)] = (unsigned char)byte;
    }
    __syncthreads();
    unsigned char* O8 = blk.q8 + (size_t)bz * Nq * blk.ld8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = lane + 64 * i;
      const int rl = c >> 3, cc = c & 7;
      if (q0 + rl < Nq) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(smem + SP_TILE_BYTES + (size_t)(wave * QW + rl) * OST_ROW + cc * 16);
        *reinterpret_cast<u32x4*>(O8 + (size_t)(q0 + rl) * blk.ld8 + hoff + cc * 16) = v;
      }
    }
  } else {
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = lane + 64 * i;
    const int rl = c >> 4, cc = c & 15;
    if (q0 + rl < Nq) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(smem + (TWO_SEG ? SP_TILE_BYTES : 0) + (size_t)(wave * QW + rl) * OST_ROW + cc * 16);
      *reinterpret_cast<u32x4*>(O + (size_t)(q0row + rl) * ldo + hoff + cc * 8) = v;
    }
  }
  }
  __syncthreads();  // the next item's tile staging overwrites the O staging area
  }  // item
}


// ------------------------------------------------------------------------------------------------
// "w4" (round 4): the V^T / LDS-DMA form with ONE WAVE PER SIMD.  4 waves x 64 query rows (two 32-row sub-blocks A, B per wave), the
// whole 512-entry register file per wave.  Same LDS images, DMA source swizzles, fragment assignments, products, rounding points and
// speculative softmax as attn_fwd_sp_kernel<false, true> (every line of arithmetic below has its twin there; the outputs are equal bit
// for bit) - what changes is who shares what, and when:
//   * a K (V^T) fragment read from LDS feeds TWO MFMAs (sub-block A, then B): 128 KiB of fragment reads per key tile and CU instead
//     of 256 KiB, and half the ds_read issue slots per MFMA;
//   * no second wave on the SIMD: the matrix pipe of a SIMD sees one in-order stream - S(t) of both sub-blocks (32 MFMAs), then
//     P(t-1).V(t-1) of both (32 MFMAs, each followed by its share of softmax(t): 2 v_exp, 2 adds, one packed conversion);
//   * nothing else covers a wave's stalls, so no phase starts with an LDS round trip and the workgroup barrier does not sit in front
//     of a phase: the first V^T fragments of P.V are read under the last S units, the first K fragments of the NEXT tile under the
//     last P.V units, and the tile's one barrier (counted wait: K(t+1) has landed) sits in the MIDDLE of the P.V phase.  The K DMA runs
//     two tiles ahead through FOUR buffers (a fast wave issues K(t+2) into the buffer of K(t-2) behind barrier t-1, which every wave
//     reaches after its last read of K(t-2) - the exact route at the end of tile t-2) and the V^T DMA of tile t+1 moves behind the
//     barrier too (its buffer held V(t-2), read until the end of tile t-1);
//   * 4 waves stage what 8 did: 4 K pieces + 4 V^T pieces of 1 KiB per wave and tile;
//   * a wave whose second sub-block has no query rows (the last block of a sequence) runs the NSB = 1 form: half the MFMAs per tile.
// Registers: O^T 2 x 4 x 16 = 128 accumulators and Q 2 x 8 x 4 = 64 in the accumulator file; S 64, P 32, offsets 32, fragment rings 40.
// ------------------------------------------------------------------------------------------------
// Matrix instructions of the w4 kernel as asm: the operand FILES are part of the design - Q fragments (64 registers) and the O^T
// accumulators (128) live in the accumulator half of the register file ("a"), S / P / the offsets / the fragment rings in the VGPR
// half, which is all the VALU can address.  Left to the builtins hipcc allocated 256 + 256 registers, spilled 93 and moved ~130
// registers between the halves at the top of every tile.  Hazards the compiler does not pad for an asm statement (guide section 5.7):
//   * D of an asm MFMA -> VALU reader: S is first read (v_exp) behind >= 3 later MFMAs; the rare readers straight behind the S phase
//     (tail mask, exact offset) and the readers of O (rescale, epilogue) sit behind W4A_SETTLE_* (s_nop 11 = 12 states, 8-pass XDL);
//   * VALU / v_accvgpr_write -> MFMA operand (2 states): P is written >= 4 units ahead of the MFMA that reads it; a rescaled O passes
//     W4A_SETTLE_O before the next P.V MFMA.
#define W4A_S_FIRST(ST, KF, QF, CI) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(ST) : "v"(KF), "a"(QF), "v"(CI))
#define W4A_S(ST, KF, QF) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ST) : "v"(KF), "a"(QF))
#define W4A_PV(OA, VF, PP) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(OA) : "v"(VF), "v"(PP))
#define W4A_ADD(ACC, X) asm volatile("v_add_f32 %0, %0, %1" : "+v"(ACC) : "v"(X))
#define W4A_CVT(DST, A, B)                                                                  \
  do {                                                                                      \
    uint32_t w_;                                                                            \
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w_) : "v"(A), "v"(B));               \
    (DST) = w_;                                                                             \
  } while (0)
#ifndef W4_KDMA_IN_PV
#define W4_KDMA_IN_PV 1
#endif
#ifndef W4_ABLATE
#define W4_ABLATE 0  // diagnostic builds only (tools/attn_body_ab.py): 1 no tile barrier, 2 no DMA, 4 no softmax fillers, 8 no K reads, 16 no V reads, 32 no S MFMAs, 64 no P.V MFMAs
#endif
#ifndef W4_EXP_ASM
#define W4_EXP_ASM 1
#endif
#if W4_EXP_ASM
#define W4A_EXP(X) asm volatile("v_exp_f32 %0, %0" : "+v"(X))
#else
#define W4A_EXP(X) (X) = __builtin_amdgcn_exp2f(X)
#endif
#define W4A_SETTLE_S(SB) asm volatile("s_nop 11" : "+v"(st[SB][0]), "+v"(st[SB][1]))
#define W4A_SETTLE_O(SB) asm volatile("s_nop 11" : "+a"(oacc[SB][0]), "+a"(oacc[SB][1]), "+a"(oacc[SB][2]), "+a"(oacc[SB][3]))
constexpr int W4_KB = 4;                       // K buffers: the DMA runs TWO tiles ahead (PMC of the one-ahead form: a fifth of the wave cycles parked
                                               // in front of the barrier - 0.8 us between issue and need is less than a loaded L2 / HBM round trip)
constexpr int W4_V0 = W4_KB * PK_TILE;         // then three V^T buffers
constexpr int W4_SMEM = W4_V0 + 3 * PV_TILE;   // 124928 B (also holds the 69632-B O staging)
template <int N>
struct w4_int {
  static constexpr int value = N;
};

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void attn_fwd_w4_kernel(
    const bf16* __restrict__ Q_, bf16* __restrict__ O_, KVSeg seg0_, int Nq, int H, int ldq, int ldo, int nqb, float scale_log2e, int batch,
    BlkRows blk) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NWAVE = 4, QWW = 64, QB = QWW * NWAVE;  // 256 query rows per workgroup, as in the 8-wave kernel: the work items are the same
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  for (int item = blockIdx.x; item < nqb * H * batch; item += gridDim.x) {
  const bf16* Q = Q_;
  bf16* O = O_;
  KVSeg sg = seg0_;
  int head, qb, bz;
  {  // work order: attn_fwd_sp_kernel (full query blocks of an XCD's heads first, the remainder blocks last)
    const int nqb_full = Nq / QB;
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
  if (blk.rows > 0) {
    Q += (size_t)bz * blk.rows * ldq;
    O += (size_t)bz * blk.rows * ldo;
    sg.k += (size_t)bz * blk.rows * sg.ldk;
    sg.v += (size_t)bz * blk.vt_cols;
  } else {
    Q += (size_t)bz * Nq * ldq;
    O += (size_t)bz * Nq * ldo;
    sg.k += (size_t)bz * sg.len * sg.ldk;
    sg.v += (size_t)bz * sg.len;
  }
  const int q0 = qb * QB + wave * QWW;
  const int hoff = head * HD;
  const bool active = q0 < Nq;
  const bool two_sub = q0 + 32 < Nq;  // (wave-uniform) the second sub-block has query rows
  const int q0row = blk.rows > 0 ? (q0 / blk.rows) * blk.stride + q0 % blk.rows : q0;  // (a wave's 64 tokens never straddle a block)

  // fragment read addresses (see attn_fwd_sp_kernel)
  const int k_w = l31 & 3;
  const unsigned char* k_rd = smem + (l31 >> 2) * K_GRP + k_w * 256 + ((hh ^ (k_w & 1)) << 4);
  const int k_eo[2] = {(k_w >> 1) << 5, ((k_w >> 1) ^ 1) << 5};
  int vt_off[4];
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) {
    const int x = (l31 >> 1) & 7;
    vt_off[s4] = l31 * 128 + ((2 * (s4 ^ (x >> 1)) + (hh ^ (x & 1))) << 4);
  }
  const int ntiles = (sg.len + KVB - 1) / KVB;
  const int k_rows_span = blk.rows > 0 ? ((sg.len - 1) / blk.rows) * blk.stride + blk.rows : sg.len;
  const auto k_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(sg.k + hoff), 0, (k_rows_span - 1) * sg.ldk * 2 + HD * 2, 0x00020000);
  const auto v_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(sg.v + (size_t)hoff * sg.ldv), 0, ((HD - 1) * sg.ldv + ntiles * KVB) * 2, 0x00020000);
  const int k_tile_bytes = KVB * sg.ldk * 2;
  // K image group g = wave + 4 j (j = 0..3) holds image rows 4 g + (lane >> 4); image row rho holds key rho with bits 2 and 3 swapped
  const int kd_row = (lane >> 4) + 8 * (wave & 1) + 4 * (wave >> 1);
  const int kd_voff0 = kd_row * sg.ldk * 2 + (((lane & 15) ^ (lane >> 4)) << 4), kd_step = 16 * sg.ldk * 2;
  // V^T piece wave + 4 j: rows 8 (wave + 4 j) + (lane >> 3), slot lane & 7 <- chunk slot ^ ((row >> 1) & 7)
  const int vd_row = 8 * wave + (lane >> 3);
  const int vd_voff0 = vd_row * sg.ldv * 2 + (((lane & 7) ^ ((vd_row >> 1) & 7)) << 4), vd_step = 32 * sg.ldv * 2;
  auto dma_v = [&](int t, int buf, int j) __attribute__((always_inline)) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(v_rsrc, (lds_void*)(smem + W4_V0 + buf * PV_TILE + (wave + 4 * j) * 1024), 16, vd_voff0 + j * vd_step,
                                             t * (KVB * 2), 0, 0);
  };
  auto dma_k = [&](int t, int buf, int j) __attribute__((always_inline)) {
    int soff = t * k_tile_bytes;
    if (blk.rows > 0) {
      const int b_ = blk.magic ? (int)__umulhi((uint32_t)t, blk.magic) : t;
      soff = (b_ * blk.stride + (t - b_ * (blk.rows >> 6)) * KVB) * sg.ldk * 2;
    }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(k_rsrc, (lds_void*)(smem + buf * PK_TILE + (wave + 4 * j) * K_GRP), 16, kd_voff0 + j * kd_step, soff, 0, 0);
  };

  // ---- prologue, every wave: tile 0 -> K buffer 0 / V buffer 0; V buffer 2 plays "V(-1)" (zeros: P(-1) = 0 must not meet NaNs)
#pragma unroll
  for (int j = 0; j < 4; ++j) dma_k(0, 0, j);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int j = 0; j < 4; ++j) dma_v(0, 0, j);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int j = 0; j < 4; ++j) dma_k(1, 1, j);
  {
    const pp_u4 z = {0u, 0u, 0u, 0u};
    for (int i = tid; i < PV_TILE / 16; i += NWAVE * 64) *reinterpret_cast<pp_u4*>(smem + W4_V0 + 2 * PV_TILE + i * 16) = z;
  }

  // Tile t of an ACTIVE wave (K buffer of tile t: t & 3; vb_prev / vb_cur = V^T buffers of tiles t - 1 / t):
  //     S(t): 16 units of NSB MFMAs on the K fragment ring (first four read under the previous P.V tail); units 0..3 issue the K DMA of
  //           tile t + 2; units 10..15 read the first six V^T(t-1) fragments
  //     tail mask (last tile) / exact offset (first tile) / O rescale (rare)
  //     P(t-1).V(t-1): units 0..7 | vmcnt(4), s_barrier: K(t+1), V(t) in LDS, V(t-2) free | units 8..15, carrying the V^T DMA of tile t + 1
  //           (8..11) and the first four K(t+1) fragment reads (12..15)
  //     row-sum check of the speculative softmax; exact route (S(t) again from kc) if it fails
  // A wave without query rows runs the DMA issues and the barriers only.
  // Vector-memory queue of a wave, oldest first, at the barrier of tile t: K(t+1) x4 (issued in tile t-1), V(t) x4 (tile t-1, behind its
  // barrier), K(t+2) x4 (tile t): vmcnt(4) = K(t+1) and V(t) have landed, K(t+2) may still fly.
  if (!active) {
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    CE_EPOCH_BARRIER();
    int vb_prev = 2, vb_cur = 0;
    for (int t = 0; t < ntiles; ++t) {
#pragma unroll
      for (int j = 0; j < 4; ++j) dma_k(t + 2, (t + 2) & 3, j);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const int vb_next = 3 - vb_prev - vb_cur;
#pragma unroll
      for (int j = 0; j < 4; ++j) dma_v(t + 1, vb_next, j);
      vb_prev = vb_cur;
      vb_cur = vb_next;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    CE_EPOCH_BARRIER();
  }
  auto run = [&](auto nsb_c) __attribute__((always_inline)) {
    constexpr int NSB = decltype(nsb_c)::value;
    bf16x8 qf[NSB][8];
#pragma unroll
    for (int sb = 0; sb < NSB; ++sb) {
      const bf16* qrow = Q + (size_t)(q0 + 32 * sb + l31 < Nq ? q0row + 32 * sb + l31 : 0) * ldq + hoff + 8 * hh;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) qf[sb][ks] = *reinterpret_cast<const bf16x8*>(qrow + 16 * ks);
    }
    f32x16 oacc[NSB][4];
#pragma unroll
    for (int sb = 0; sb < NSB; ++sb)
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[sb][m][r] = 0.f;
    float l_run[NSB], alpha_prev[NSB], mc[NSB];
    f32x16 cinit[NSB];
    u32x4 ppk[NSB][4];
#pragma unroll
    for (int sb = 0; sb < NSB; ++sb) {
      l_run[sb] = 0.f;
      alpha_prev[sb] = 1.0f;
      mc[sb] = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) cinit[sb][r] = 0.f;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) ppk[sb][s4] = u32x4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int sb = 0; sb < NSB; ++sb) {  // Q scaled once by softmax_scale * log2(e), rounded back to bf16 (see attn_fwd_sp_kernel)
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const u32x4 w = __builtin_bit_cast(u32x4, qf[sb][ks]);
        u32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = pack_bf16(bf16lo(w[i]) * scale_log2e, bf16hi(w[i]) * scale_log2e);
        qf[sb][ks] = __builtin_bit_cast(bf16x8, o);
      }
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) asm volatile("" : "+a"(qf[sb][ks]));  // retire the Q loads before the loop; Q lives in the accumulator file
    }
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // (the Q loads above already drained the queue; K(1) may fly in principle)
    CE_EPOCH_BARRIER();  // K(0), V(0) and the zeros of "V(-1)" are in LDS, for everybody

#define CE_LDK(KB, i) (*reinterpret_cast<const bf16x8*>((KB) + k_eo[((i) >> 1) & 1] + ((i) & 1) * 8 * K_GRP + ((i) >> 2) * 64))
#define CE_LDV(VB, u) (*reinterpret_cast<const bf16x8*>((VB) + vt_off[(u) >> 2] + ((u) & 3) * 4096))
    constexpr int RING = 4, VRING = 6;
    bf16x8 kf[RING], vf[VRING];
#pragma unroll
    for (int i = 0; i < RING; ++i) kf[i] = CE_LDK(k_rd, i);
    int vb_prev = 2, vb_cur = 0;
    for (int t = 0; t < ntiles; ++t) {
      const int vb_next = 3 - vb_prev - vb_cur;
      const unsigned char* kb = k_rd + (t & 3) * PK_TILE;
      const unsigned char* kbn = k_rd + ((t + 1) & 3) * PK_TILE;
      const unsigned char* vtb = smem + W4_V0 + vb_prev * PV_TILE;  // V^T image of tile t-1
      const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      f32x16 st[NSB][2];
      // ---- S^T(t) = K(t).Q^T: fragment i (kv fragment f = i & 1, k-step ks = i >> 1) feeds every sub-block; one scheduling region each
#pragma unroll
      for (int i = 0; i < 16; ++i) {
#pragma unroll
        for (int sb = 0; sb < NSB; ++sb) {
          if (i < 2) W4A_S_FIRST(st[sb][i & 1], kf[i % RING], qf[sb][i >> 1], cinit[sb]);  // untied form: the chain starts from the offset registers
          else if (!(W4_ABLATE & 32)) W4A_S(st[sb][i & 1], kf[i % RING], qf[sb][i >> 1]);
        }
        if (i + RING < 16 && !(W4_ABLATE & 8)) kf[i % RING] = CE_LDK(kb, i + RING);
#if !W4_KDMA_IN_PV
        if (i < 4 && !(W4_ABLATE & 2)) dma_k(t + 2, (t + 2) & 3, i);
#endif
        if (i >= 16 - VRING && !(W4_ABLATE & 16)) vf[i - (16 - VRING)] = CE_LDV(vtb, i - (16 - VRING));
        __builtin_amdgcn_sched_barrier(0);
      }
      float alpha[NSB], psum[NSB];
#pragma unroll
      for (int sb = 0; sb < NSB; ++sb) {
        alpha[sb] = 1.0f;
        psum[sb] = 0.f;
      }
      auto mask_tail = [&](int sb) __attribute__((always_inline)) {
        if ((t + 1) * KVB > sg.len) {
          W4A_SETTLE_S(sb);
          int thr = sg.len - t * KVB - 8 * hh;  // (opaque inside the branch: see attn_fwd_sp_kernel - the block must not be if-converted)
          asm volatile("" : "+v"(thr));
#pragma unroll
          for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int c = 32 * f + 16 * (r >> 3) + 4 * ((r >> 2) & 1) + (r & 3);  // pi(row), see dma_k
              if (c >= thr) st[sb][f][r] = NEG_BIG;
            }
        }
      };
      auto rebase = [&](int sb) __attribute__((always_inline)) {  // exact tile max of both halves of the row: move the offset
        W4A_SETTLE_S(sb);
        float mx;
        {
          float m0 = fmaxf(st[sb][0][0], st[sb][0][1]), m1 = fmaxf(st[sb][0][8], st[sb][0][9]), m2 = fmaxf(st[sb][1][0], st[sb][1][1]),
                m3 = fmaxf(st[sb][1][8], st[sb][1][9]);
#pragma unroll
          for (int r = 2; r < 8; ++r) {
            m0 = fmaxf(m0, st[sb][0][r]);
            m1 = fmaxf(m1, st[sb][0][8 + r]);
            m2 = fmaxf(m2, st[sb][1][r]);
            m3 = fmaxf(m3, st[sb][1][8 + r]);
          }
          mx = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
        }
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
        mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        const float shift = t == 0 ? mx : fmaxf(mx, 0.f);
        alpha[sb] = __builtin_amdgcn_exp2f(-shift);
        mc[sb] += shift;
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
          for (int r = 0; r < 16; ++r) st[sb][f][r] -= shift;
#pragma unroll
        for (int r = 0; r < 16; ++r) cinit[sb][r] = -mc[sb];
      };
      auto pack_pair = [&](int sb, int j) __attribute__((always_inline)) {
        const f32x2 pr = {st[sb][j >> 3][(2 * j) & 15], st[sb][j >> 3][(2 * j + 1) & 15]};
        ppk[sb][j >> 2][j & 3] = __builtin_bit_cast(uint32_t, __builtin_convertvector(pr, bf16x2));
      };
#pragma unroll
      for (int sb = 0; sb < NSB; ++sb) {
        mask_tail(sb);
        if (t == 0) rebase(sb);
        if (__any(alpha_prev[sb] != 1.0f)) {  // O at the scale of m(t-1) before P(t-1).V(t-1) is added (rare after the first tiles)
          W4A_SETTLE_O(sb);
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[sb][m][r] *= alpha_prev[sb];
          W4A_SETTLE_O(sb);
        }
      }
      // ---- O^T += V^T(t-1).P^T(t-1) on the matrix pipe, P(t) on the VALU: unit u = (k-step u >> 2, dv fragment u & 3) = one V^T fragment,
      // one MFMA per sub-block, and per sub-block the v_exp of elements 2u, 2u+1, two adds into the row sum (one unit behind) and one
      // packed conversion (four units behind: P(t) replaces P(t-1) in place).  One scheduling region per MFMA.
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        if (u == 8) {  // K(t+1) and V(t) have landed (this wave's pieces; everybody's behind the barrier); all reads of V(t-2), K(t-1) are over
          asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
          if (!(W4_ABLATE & 1)) __builtin_amdgcn_s_barrier();
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int sb = 0; sb < NSB; ++sb) {
          if (!(W4_ABLATE & 64)) W4A_PV(oacc[sb][u & 3], vf[u % VRING], ppk[sb][u >> 2]);
          if (sb == NSB - 1) {
            if (u + VRING < 16 && !(W4_ABLATE & 16)) vf[u % VRING] = CE_LDV(vtb, u + VRING);
#if W4_KDMA_IN_PV
            // the K DMA of tile t + 2 rides in the P.V phase too: an LDS-DMA piece costs its wave 100-185 cycles of issue among the ds_reads
            // of the S phase and 25-60 in VALU-only gaps (MI355X guide, cycle-constant table) - with one wave per SIMD nobody covers that
            if (u < 4 && !(W4_ABLATE & 2)) dma_k(t + 2, (t + 2) & 3, u);
#endif
            if (u >= 8 && u < 12 && !(W4_ABLATE & 2)) dma_v(t + 1, vb_next, u - 8);
            if (u >= 12 && !(W4_ABLATE & 8)) kf[u - 12] = CE_LDK(kbn, u - 12);
          }
          // hand-placed: add (previous unit's first element), exp, add, exp, pack - adds as asm (as plain C the two sub-blocks' sums were
          // SLP-packed into one dependent v_pk_add_f32 chain), exps as asm so that they sit BETWEEN the two dependent adds (hipcc pads a
          // wait state between two back-to-back asm statements on one register)
          // (the conversion as asm too: as a builtin hipcc SANK the second sub-block's sixteen conversions out of this block, behind the
          // last MFMA of the tile - nothing consumes P(t) before the next tile)
#define W4_F_ADD0 if (u > 0) W4A_ADD(psum[sb], st[sb][(u - 1) >> 3][(2 * u - 2) & 15])
#define W4_F_ADD1 if (u > 0) W4A_ADD(psum[sb], st[sb][(u - 1) >> 3][(2 * u - 1) & 15])
#define W4_F_CVT if (u >= 4) W4A_CVT(ppk[sb][(u - 4) >> 2][(u - 4) & 3], st[sb][(u - 4) >> 3][(2 * (u - 4)) & 15], st[sb][(u - 4) >> 3][(2 * (u - 4) + 1) & 15])
#define W4_F_EXP0 W4A_EXP(st[sb][(2 * u) >> 4][(2 * u) & 15])
#define W4_F_EXP1 W4A_EXP(st[sb][(2 * u + 1) >> 4][(2 * u + 1) & 15])
#ifndef W4_ORDER
#define W4_ORDER 0
#endif
#if W4_ABLATE & 4
#elif W4_ORDER == 0
          W4_F_ADD0; W4_F_CVT; W4_F_EXP0; W4_F_ADD1; W4_F_EXP1;
#elif W4_ORDER == 1
          W4_F_ADD0; W4_F_EXP0; W4_F_CVT; W4_F_ADD1; W4_F_EXP1;
#elif W4_ORDER == 2
          W4_F_CVT; W4_F_ADD0; W4_F_EXP0; W4_F_EXP1; W4_F_ADD1;
#elif W4_ORDER == 3
          W4_F_EXP0; W4_F_EXP1; W4_F_ADD0; W4_F_CVT; W4_F_ADD1;
#elif W4_ORDER == 4
          W4_F_EXP0; W4_F_ADD0; W4_F_EXP1; W4_F_CVT; W4_F_ADD1;
#elif W4_ORDER == 5
          W4_F_ADD0; W4_F_EXP0; W4_F_EXP1; W4_F_CVT; W4_F_ADD1;
#endif
#undef W4_F_ADD0
#undef W4_F_ADD1
#undef W4_F_CVT
#undef W4_F_EXP0
#undef W4_F_EXP1
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int sb = 0; sb < NSB; ++sb) {
        W4A_ADD(psum[sb], st[sb][1][14]);
        W4A_ADD(psum[sb], st[sb][1][15]);
#pragma unroll
        for (int j = 12; j < 16; ++j) pack_pair(sb, j);
      }
#pragma unroll
      for (int sb = 0; sb < NSB; ++sb) {
        if (__builtin_expect(W4_ABLATE == 0 && t > 0 && __any(psum[sb] > SP_SPEC_THR), 0)) {
          CE_DIAG_COUNT_EXACT(g_sp_exact_hits);
          // exact_tile: S(t) again (K(t) stays in its buffer until barrier t + 1), true row max, plain softmax
          st[sb][0] = zero16;
          st[sb][1] = zero16;
          asm volatile("s_nop 1" : "+v"(st[sb][0]), "+v"(st[sb][1]));  // VALU write -> MFMA operand
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const bf16x8 kfr = CE_LDK(kb, i);
            W4A_S(st[sb][i & 1], kfr, qf[sb][i >> 1]);
          }
          W4A_SETTLE_S(sb);
#pragma unroll
          for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) st[sb][f][r] -= mc[sb];
          mask_tail(sb);
          rebase(sb);
          psum[sb] = 0.f;
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            st[sb][j >> 3][(2 * j) & 15] = __builtin_amdgcn_exp2f(st[sb][j >> 3][(2 * j) & 15]);
            st[sb][j >> 3][(2 * j + 1) & 15] = __builtin_amdgcn_exp2f(st[sb][j >> 3][(2 * j + 1) & 15]);
            psum[sb] += st[sb][j >> 3][(2 * j) & 15] + st[sb][j >> 3][(2 * j + 1) & 15];
            pack_pair(sb, j);
          }
        }
        l_run[sb] = l_run[sb] * alpha[sb] + psum[sb];
        alpha_prev[sb] = alpha[sb];
      }
      vb_prev = vb_cur;
      vb_cur = vb_next;
    }
    // ---- drain: P(ntiles-1).V(ntiles-1) (its V buffer, now vb_prev, became visible at the last barrier); the surplus prefetches of
    // the last tile must retire before the O staging overlays the tile buffers
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    {
      const unsigned char* vtb = smem + W4_V0 + vb_prev * PV_TILE;
#pragma unroll
      for (int sb = 0; sb < NSB; ++sb)
        if (__any(alpha_prev[sb] != 1.0f)) {
          W4A_SETTLE_O(sb);
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[sb][m][r] *= alpha_prev[sb];
          W4A_SETTLE_O(sb);
        }
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const bf16x8 vfd = CE_LDV(vtb, u);
#pragma unroll
        for (int sb = 0; sb < NSB; ++sb) W4A_PV(oacc[sb][u & 3], vfd, ppk[sb][u >> 2]);
      }
    }
#undef CE_LDK
#undef CE_LDV
#pragma unroll
    for (int sb = 0; sb < NSB; ++sb) W4A_SETTLE_O(sb);
    CE_EPOCH_BARRIER();  // every wave is done with the tile buffers (the O staging overlays them)
#pragma unroll
    for (int sb = 0; sb < NSB; ++sb) {
      unsigned char* ost = smem + (size_t)(wave * QWW + 32 * sb + l31) * OST_ROW;
      const float l_tot = l_run[sb] + __shfl_xor(l_run[sb], 32, 64);
      const float inv = 1.0f / l_tot;
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const uint32_t w0 = pack_bf16(oacc[sb][m][4 * a + 0] * inv, oacc[sb][m][4 * a + 1] * inv);
          const uint32_t w1 = pack_bf16(oacc[sb][m][4 * a + 2] * inv, oacc[sb][m][4 * a + 3] * inv);
          *reinterpret_cast<u32x2*>(ost + (32 * m + 8 * a + 4 * hh) * 2) = u32x2{w0, w1};
        }
    }
  };
  if (active) {
    if (two_sub) run(w4_int<2>{});
    else run(w4_int<1>{});
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {  // the wave's 64 staged rows as whole 256-B rows, 16 B per lane
    const int c = lane + 64 * i;
    const int rl = c >> 4, cc = c & 15;
    if (q0 + rl < Nq) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(smem + (size_t)(wave * QWW + rl) * OST_ROW + cc * 16);
      *reinterpret_cast<u32x4*>(O + (size_t)(q0row + rl) * ldo + hoff + cc * 8) = v;
    }
  }
  __syncthreads();  // the next item's tile staging overwrites the O staging area
  }  // item
}
#undef W4A_S_FIRST
#undef W4A_S
#undef W4A_PV
#undef W4A_ADD
#undef W4A_CVT
#undef W4A_EXP
#undef W4A_SETTLE_S
#undef W4A_SETTLE_O

// ------------------------------------------------------------------------------------------------
// V [rows = keys of all samples][ldv] (head h at columns 128 h ..) -> V^T [H * 128][ldvt] (row = head channel, column = key; the
// samples' keys side by side), the operand layout of the VT form of the attention kernel above.  One workgroup per 64-key strip
// and head: 16-B loads along the channels, 4 x 4 patches transposed in registers, LDS for the change of the fast axis, 16-B stores
// along the keys.  Columns [n_keys, ldvt) are zeroed by the strip that owns them (they meet P = 0 and must be finite).
// ------------------------------------------------------------------------------------------------
// Blocked input (blk_rows > 0, the all-to-all receive layout: key g of sample b = blockIdx.z in row (g / blk_rows) blk_stride + b blk_rows
// + g % blk_rows; a 64-key strip never straddles a block): the output is plain per sample, sample b at columns [b vt_cols, ...).
__global__ __launch_bounds__(256) void v_transpose_kernel(const bf16* __restrict__ V, int ldv, bf16* __restrict__ VT, int ldvt, int n_keys,
                                                          int blk_rows, int blk_stride, int vt_cols) {
  __shared__ __attribute__((aligned(16))) unsigned char tile[HD * (KVB * 2 + 16)];  // [128 d][64 keys] bf16, rows 144 B
  constexpr int TROW = KVB * 2 + 16;
  const int tid = threadIdx.x, head = blockIdx.y, key0 = blockIdx.x * KVB;
  if (blk_rows > 0) {
    V += ((size_t)(key0 / blk_rows) * blk_stride + (size_t)blockIdx.z * blk_rows + key0 % blk_rows - key0) * ldv;  // row of key0, minus key0
    VT += (size_t)blockIdx.z * vt_cols;
  }
  const int col_end = blk_rows > 0 ? vt_cols : ldvt;  // columns this sample owns
  // thread -> patch (4 keys x 4 channels): kq = tid & 15 (key quad), dq = tid >> 4 (channel quad, + 16 per pass)
  const int kq = tid & 15, dq = tid >> 4;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int d0 = 4 * (dq + 16 * pass);
    pp_u2 r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int key = key0 + 4 * kq + i;
      r[i] = key < n_keys ? *reinterpret_cast<const pp_u2*>(V + (size_t)key * ldv + head * HD + d0) : pp_u2{0u, 0u};
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {  // channel d0 + j of keys 4 kq .. 4 kq + 3
      const int w = j >> 1;
      uint32_t lo, hi;
      if ((j & 1) == 0) {
        lo = (r[0][w] & 0xffffu) | (r[1][w] << 16);
        hi = (r[2][w] & 0xffffu) | (r[3][w] << 16);
      } else {
        lo = (r[0][w] >> 16) | (r[1][w] & 0xffff0000u);
        hi = (r[2][w] >> 16) | (r[3][w] & 0xffff0000u);
      }
      *reinterpret_cast<pp_u2*>(tile + (d0 + j) * TROW + kq * 8) = pp_u2{lo, hi};
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 256 * i;  // 1024 chunks of 16 B: row d = c >> 3, chunk c & 7
    const int d = c >> 3, ch = c & 7;
    const pp_u4 v = *reinterpret_cast<const pp_u4*>(tile + d * TROW + ch * 16);
    const int col = key0 + 8 * ch;
    if (col < col_end) *reinterpret_cast<pp_u4*>(VT + (size_t)(head * HD + d) * ldvt + col) = v;
  }
}

}  // namespace

// Kernel selection (host-side knob): 0 = automatic = 64 = the software-pipelined kernel on the padded LDS images ("sp");
// 4 / 8 = the plain kernel with 4 / 8 waves per workgroup (the readable statement of the algorithm; kept as the A/B partner
// of the parity tests).  The other loop bodies of round 1 (XOR-swizzled pipeline, ping-pong, one wave per SIMD) and their
// ablation build lost every A/B and were removed in round 2; git history has them.
#ifdef CE_DIAGNOSTICS
extern "C" int ce_diag_mx_exact_hits(unsigned long long* out, int reset);
// hits[0] = waves x key tiles that took the exact route in the bf16 software-pipelined kernels (attn_fwd_sp_kernel, attn_fwd_w4_kernel) since
// the last reset, hits[1] = the same for the MXFP8 kernel; reset != 0 zeroes both.  Synchronises the device (a diagnostic, not a launcher).
CE_API int ce_diag_attention_exact_route_hits(unsigned long long* hits, int reset) {
  if (!hits) return CE_ERR_ARG;
  unsigned long long v = 0ull, z = 0ull;
  if (hipDeviceSynchronize() != hipSuccess) return CE_ERR_ARG;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_sp_exact_hits), sizeof(v)) != hipSuccess) return CE_ERR_ARG;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_sp_exact_hits), &z, sizeof(z)) != hipSuccess) return CE_ERR_ARG;
  hits[0] = v;
  return ce_diag_mx_exact_hits(hits + 1, reset);
}
#endif
CE_KNOB g_attn_nwave = 0;
#ifdef CE_DIAGNOSTICS
CE_API int ce_set_attention_waves(int nwave) {
  const int old = g_attn_nwave;
  if (nwave == 0 || nwave == 4 || nwave == 8 || nwave == 16 || nwave == 64 || nwave == 128 || nwave == 129) g_attn_nwave = nwave;
  return old;
}
#endif

// Q [Nq][ldq], K*/V* [len][ld*], O [Nq][ldo]; all bf16, head h occupies columns [128 h, 128 h + 128).
// Second kv segment optional (k2 == nullptr or len2 == 0): O = bf16(attn(seg1)) + bf16(attn(seg2)).
// batch > 1: `batch` samples stacked along the rows of every operand (sample b: rows [b Nq, (b+1) Nq) of Q/O, rows
// [b len, (b+1) len) of each K/V segment), one launch - more workgroups per launch, smaller last-round tail.
CE_API int ce_attention_batched_bf16(const void* Q, const void* K1, const void* V1, int len1, int ldk1, int ldv1,
                                         const void* K2, const void* V2, int len2, int ldk2, int ldv2, void* O, int Nq, int H,
                                         int head_dim, int ldq, int ldo, float softmax_scale, int batch, hipStream_t stream) {
  if (!Q || !K1 || !V1 || !O) return CE_ERR_ARG;
  if (head_dim != HD || Nq <= 0 || H <= 0 || len1 <= 0 || batch <= 0 || batch > 65535) return CE_ERR_SHAPE;
  if ((ldq & 7) || (ldo & 7) || (ldk1 & 7) || (ldv1 & 3)) return CE_ERR_ALIGN;
  const bool two = (K2 != nullptr && V2 != nullptr && len2 > 0);
  if (two && ((ldk2 & 7) || (ldv2 & 3))) return CE_ERR_ALIGN;
  KVSeg s0{(const bf16*)K1, (const bf16*)V1, len1, ldk1, ldv1};
  KVSeg s1{(const bf16*)K2, (const bf16*)V2, two ? len2 : 0, ldk2, ldv2};
  const float sl2 = softmax_scale * 1.4426950408889634f;
  const bool sp = g_attn_nwave == 64 || g_attn_nwave == 0 || g_attn_nwave >= 128;  // (128 / 129 select a loop body of the V^T form only)
  const int nwave = sp ? 8 : g_attn_nwave;
  const int nqb = (Nq + nwave * QW - 1) / (nwave * QW);
  dim3 grid(nqb * H, batch), block(nwave * 64);
  if (sp) {
#define CE_ATTN_SP(TWO)                                                                                              \
  do {                                                                                                               \
    static bool done_[CE_MAX_DEVICES] = {};                                                                          \
    bool& done = done_[ce_device_slot()];                                                                            \
    if (!done) {                                                                                                     \
      (void)hipFuncSetAttribute((const void*)attn_fwd_sp_kernel<TWO>, hipFuncAttributeMaxDynamicSharedMemorySize,    \
                                sp_smem_bytes(TWO));                                                                 \
      done = true;                                                                                                   \
    }                                                                                                                \
    hipLaunchKernelGGL((attn_fwd_sp_kernel<TWO>), dim3(nqb * H * batch), block, sp_smem_bytes(TWO), stream, (const bf16*)Q, \
                       (bf16*)O, s0, s1, Nq, H, ldq, ldo, nqb, sl2, batch, BlkRows{0, 0, 0u, 0});                    \
  } while (0)
    if (two) CE_ATTN_SP(true); else CE_ATTN_SP(false);
#undef CE_ATTN_SP
    return (int)hipGetLastError();
  }
#define CE_ATTN_LAUNCH(TWO, NW)                                                                                     \
  do {                                                                                                              \
    static bool done_[CE_MAX_DEVICES] = {};                                                                         \
    bool& done = done_[ce_device_slot()];                                                                           \
    if (!done) {                                                                                                    \
      (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<TWO, NW>, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                smem_bytes(NW, TWO));                                                               \
      done = true;                                                                                                  \
    }                                                                                                               \
    hipLaunchKernelGGL((attn_fwd_kernel<TWO, NW>), grid, block, smem_bytes(NW, TWO), stream, (const bf16*)Q, (bf16*)O, \
                       s0, s1, Nq, H, ldq, ldo, nqb, sl2);                                                          \
  } while (0)
  if (two) {
    if (nwave == 8) CE_ATTN_LAUNCH(true, 8); else CE_ATTN_LAUNCH(true, 4);
  } else {
    if (nwave == 8) CE_ATTN_LAUNCH(false, 8); else CE_ATTN_LAUNCH(false, 4);
  }
#undef CE_ATTN_LAUNCH
  return (int)hipGetLastError();
}

CE_API int ce_attention_bf16(const void* Q, const void* K1, const void* V1, int len1, int ldk1, int ldv1, const void* K2,
                                 const void* V2, int len2, int ldk2, int ldv2, void* O, int Nq, int H, int head_dim, int ldq,
                                 int ldo, float softmax_scale, hipStream_t stream) {
  return ce_attention_batched_bf16(Q, K1, V1, len1, ldk1, ldv1, K2, V2, len2, ldk2, ldv2, O, Nq, H, head_dim, ldq, ldo,
                                   softmax_scale, 1, stream);
}

/* V [batch * n_tokens][ldv] -> V^T [H * 128][ldvt]: see v_transpose_kernel. */
CE_API int ce_v_transpose_bf16(const void* v, int ldv, void* vt, int ldvt, int n_keys, int H, hipStream_t stream) {
  if (!v || !vt) return CE_ERR_ARG;
  if (n_keys <= 0 || H <= 0 || ldvt < n_keys) return CE_ERR_SHAPE;
  if ((ldv & 3) || (ldvt & 7)) return CE_ERR_ALIGN;
  hipLaunchKernelGGL(v_transpose_kernel, dim3((ldvt + KVB - 1) / KVB, H), dim3(256), 0, stream, (const bf16*)v, ldv, (bf16*)vt, ldvt, n_keys, 0, 0, 0);
  return (int)hipGetLastError();
}

/* The same from the BLOCKED row layout (ce_attention_vt_blocked_bf16): key g of sample b in row (g / blk_rows) blk_stride + b blk_rows +
 * g % blk_rows of v; output plain per sample, sample b's n_keys keys at columns [b vt_sample_cols, ...), the rest of its vt_sample_cols
 * columns zeroed.  vt_sample_cols: a multiple of 64, >= 64 ceil(n_keys / 64); ldvt >= batch * vt_sample_cols. */
CE_API int ce_v_transpose_blocked_bf16(const void* v, int ldv, void* vt, int ldvt, int n_keys, int H, int batch, int blk_rows, int blk_stride,
                                           int vt_sample_cols, hipStream_t stream) {
  if (!v || !vt) return CE_ERR_ARG;
  if (n_keys <= 0 || H <= 0 || batch <= 0 || blk_rows <= 0 || (blk_rows & 63) || blk_stride < batch * blk_rows || (vt_sample_cols & 63) ||
      vt_sample_cols < (n_keys + KVB - 1) / KVB * KVB || ldvt < batch * vt_sample_cols)
    return CE_ERR_SHAPE;
  if ((ldv & 3) || (ldvt & 7)) return CE_ERR_ALIGN;
  hipLaunchKernelGGL(v_transpose_kernel, dim3(vt_sample_cols / KVB, H, batch), dim3(256), 0, stream, (const bf16*)v, ldv, (bf16*)vt, ldvt, n_keys,
                     blk_rows, blk_stride, vt_sample_cols);
  return (int)hipGetLastError();
}

// the 16 x 16 x 32 body of the plain-layout V^T attention (ce_attn16.hip; ce_set_attention_waves(16))
#ifdef CE_DIAGNOSTICS
extern "C" int ce_attn16_launch(const void* Q, const void* K, const void* Vt, int len, int ldk, int ldvt, void* O, int Nq, int H, int ldq, int ldo,
                                float sl2, int batch, int cus, hipStream_t stream);
#endif

/* Self-attention with V handed over TRANSPOSED (V^T [H * 128][ldvt]): both K and V^T tiles reach LDS by LDS-DMA, no register
 * staging.  One KV segment, software-pipelined kernel only.  blk_rows == 0: plain layout (sample b's token g in row b N + g of Q / K / O,
 * its keys in columns [b len, (b + 1) len) of V^T).  blk_rows > 0 (a multiple of 64): the BLOCKED layout an all-to-all leaves behind -
 * token g of sample b in row (g / blk_rows) blk_stride + b blk_rows + g % blk_rows of Q / K / O; V^T plain per sample with column
 * stride vt_sample_cols; Nq = number of query tokens per sample (a multiple of blk_rows), len = number of VALID keys. */
static int attention_vt_launch(const void* Q, const void* K, const void* Vt, int len, int ldk, int ldvt, void* O, int Nq, int H, int head_dim,
                               int ldq, int ldo, float softmax_scale, int batch, int blk_rows, int blk_stride, int vt_sample_cols,
                               hipStream_t stream) {
  if (!Q || !K || !Vt || !O) return CE_ERR_ARG;
  if (head_dim != HD || Nq <= 0 || H <= 0 || len <= 0 || batch <= 0 || batch > 65535) return CE_ERR_SHAPE;
  const int ntiles_cols = (len + KVB - 1) / KVB * KVB;
  if (blk_rows == 0) {
    vt_sample_cols = len;
    if (ldvt < (batch - 1) * len + ntiles_cols) return CE_ERR_SHAPE;  // the last tile of the last sample reads whole 64-key strips
  } else {
    if ((blk_rows & 63) || blk_stride < batch * blk_rows || (Nq % blk_rows) || vt_sample_cols < ntiles_cols ||
        ldvt < (batch - 1) * vt_sample_cols + ntiles_cols)
      return CE_ERR_SHAPE;
  }
  if ((ldq & 7) || (ldo & 7) || (ldk & 7) || (ldvt & 7) || (batch > 1 && (vt_sample_cols & 1))) return CE_ERR_ALIGN;  // sample b's columns start at byte 2 b cols: dword-aligned DMA source
  KVSeg s0{(const bf16*)K, (const bf16*)Vt, len, ldk, ldvt};
  KVSeg s1{nullptr, nullptr, 0, 0, 0};
  BlkRows blk{blk_rows, blk_stride, 0u, vt_sample_cols};
  if (blk_rows > 0) {
    const uint32_t tps = (uint32_t)(blk_rows >> 6);
    blk.magic = tps == 1 ? 0u : (uint32_t)(((1ull << 32) + tps - 1) / tps);  // exact for tile indices < 2^16
  }
  const float sl2 = softmax_scale * 1.4426950408889634f;
  const int nqb = (Nq + 8 * QW - 1) / (8 * QW);
  static bool done_[CE_MAX_DEVICES] = {};
  bool& done = done_[ce_device_slot()];
  if (!done) {
    (void)hipFuncSetAttribute((const void*)attn_fwd_sp_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, sp_smem_bytes(false));
    done = true;
  }
  static int cus_[CE_MAX_DEVICES] = {};
  int& cus = cus_[ce_device_slot()];
  if (cus == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  }
  const int items = nqb * H * batch;
#ifdef CE_DIAGNOSTICS
  if (g_attn_nwave == 16 && blk_rows == 0) {  // the 16 x 16 x 32 geometry (plain layout; unaligned operands fall through to the default body)
    const int rc = ce_attn16_launch(Q, K, Vt, len, ldk, ldvt, O, Nq, H, ldq, ldo, sl2, batch, cus, stream);
    if (rc != CE_ERR_ALIGN) return rc;
  }
#endif
  if (g_attn_nwave == 128 || g_attn_nwave == 129) {  // one wave per SIMD (attn_fwd_w4_kernel): 128 = one workgroup per item, 129 = #CUs persistent workgroups
    static bool done4_[CE_MAX_DEVICES] = {};
    bool& done4 = done4_[ce_device_slot()];
    if (!done4) {
      (void)hipFuncSetAttribute((const void*)attn_fwd_w4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, W4_SMEM);
      done4 = true;
    }
    const int grid4 = (g_attn_nwave == 129 && items > cus) ? (cus & ~7) : items;
    hipLaunchKernelGGL(attn_fwd_w4_kernel, dim3(grid4), dim3(256), W4_SMEM, stream, (const bf16*)Q, (bf16*)O, s0, Nq, H, ldq, ldo, nqb,
                       sl2, batch, blk);
    return (int)hipGetLastError();
  }
  const int grid_vt = items <= 2 * cus ? items : ((2 * cus) & ~7);  // persistent: two workgroups per CU walk the work order (see the kernel); a multiple of 8 keeps heads on their XCD
  hipLaunchKernelGGL((attn_fwd_sp_kernel<false, true>), dim3(grid_vt), dim3(512), sp_smem_bytes(false), stream, (const bf16*)Q, (bf16*)O,
                     s0, s1, Nq, H, ldq, ldo, nqb, sl2, batch, blk);
  return (int)hipGetLastError();
}

CE_API int ce_attention_vt_bf16(const void* Q, const void* K, const void* Vt, int len, int ldk, int ldvt, void* O, int Nq, int H,
                                    int head_dim, int ldq, int ldo, float softmax_scale, int batch, hipStream_t stream) {
  return attention_vt_launch(Q, K, Vt, len, ldk, ldvt, O, Nq, H, head_dim, ldq, ldo, softmax_scale, batch, 0, 0, 0, stream);
}

CE_API int ce_attention_vt_blocked_bf16(const void* Q, const void* K, const void* Vt, int len, int ldk, int ldvt, void* O, int Nq, int H,
                                            int head_dim, int ldq, int ldo, float softmax_scale, int batch, int blk_rows, int blk_stride,
                                            int vt_sample_cols, hipStream_t stream) {
  if (blk_rows <= 0) return CE_ERR_SHAPE;
  return attention_vt_launch(Q, K, Vt, len, ldk, ldvt, O, Nq, H, head_dim, ldq, ldo, softmax_scale, batch, blk_rows, blk_stride, vt_sample_cols,
                             stream);
}


/* Cross-attention (two key / value segments with a softmax each, outputs added in bf16) with both V operands handed over TRANSPOSED:
 * V1t [H * 128][ldv1t], V2t [H * 128][ldv2t], sample b's keys at columns [b vt_cols, b vt_cols + len) of its segment (vt_cols a multiple
 * of 2, >= len; a row must extend to whole 64-key strips past the LAST sample's first column: ldv*t >= (batch - 1) vt_cols + 64 ceil(len / 64);
 * columns past a sample's len - the next sample's keys or padding - only ever meet P = 0 and must be finite).  K as in
 * ce_attention_batched_bf16 (samples stacked along the rows).  K and V^T tiles of both segments go by LDS-DMA. */
static int attention_2seg_vt_launch(const void* Q, const void* K1, const void* V1t, int len1, int ldk1, int ldv1t, int vt_cols1,
                                    const void* K2, const void* V2t, int len2, int ldk2, int ldv2t, int vt_cols2, void* O, int Nq,
                                    int H, int head_dim, int ldq, int ldo, float softmax_scale, int batch, void* O8, void* S8, int ldo8,
                                    hipStream_t stream) {
  if (!Q || !K1 || !V1t || !K2 || !V2t || (!O && !O8)) return CE_ERR_ARG;
  if (head_dim != HD || Nq <= 0 || H <= 0 || len1 <= 0 || len2 <= 0 || batch <= 0 || batch > 65535) return CE_ERR_SHAPE;
  const int c1 = (len1 + KVB - 1) / KVB * KVB, c2 = (len2 + KVB - 1) / KVB * KVB;
  if (vt_cols1 < len1 || vt_cols2 < len2 || ldv1t < (batch - 1) * vt_cols1 + c1 || ldv2t < (batch - 1) * vt_cols2 + c2) return CE_ERR_SHAPE;
  if ((ldq & 7) || (ldo & 7) || (ldk1 & 7) || (ldk2 & 7) || (ldv1t & 7) || (ldv2t & 7) || (vt_cols1 & 1) || (vt_cols2 & 1)) return CE_ERR_ALIGN;
  KVSeg s0{(const bf16*)K1, (const bf16*)V1t, len1, ldk1, ldv1t};
  KVSeg s1{(const bf16*)K2, (const bf16*)V2t, len2, ldk2, ldv2t};
  const float sl2 = softmax_scale * 1.4426950408889634f;
  const int nqb = (Nq + 8 * QW - 1) / (8 * QW);
  static bool done_[CE_MAX_DEVICES] = {};
  bool& done = done_[ce_device_slot()];
  if (!done) {
    (void)hipFuncSetAttribute((const void*)attn_fwd_sp_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, sp_smem_bytes(true));
    (void)hipFuncSetAttribute((const void*)attn_fwd_sp_kernel<true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, sp_smem_bytes(true));
    done = true;
  }
  const BlkRows blk{0, vt_cols2, 0u, vt_cols1, (unsigned char*)O8, (unsigned char*)S8, ldo8};  // plain rows; the two column strides
  // persistent like the single-segment V^T launch: two workgroups per CU walk the work order (an item is 13 key tiles here: -3 % against
  // one workgroup per item, profiles/r04_cross_attention_persistent_ab.txt; 3 and 4 per CU are level with 2)
  static int cus2_[CE_MAX_DEVICES] = {};
  int& cus = cus2_[ce_device_slot()];
  if (cus == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  }
  const int items = nqb * H * batch;
  const int grid = items <= 2 * cus ? items : ((2 * cus) & ~7);
  if (O8)
    hipLaunchKernelGGL((attn_fwd_sp_kernel<true, true, true>), dim3(grid), dim3(512), sp_smem_bytes(true), stream, (const bf16*)Q,
                       (bf16*)O, s0, s1, Nq, H, ldq, ldo, nqb, sl2, batch, blk);
  else
    hipLaunchKernelGGL((attn_fwd_sp_kernel<true, true>), dim3(grid), dim3(512), sp_smem_bytes(true), stream, (const bf16*)Q, (bf16*)O,
                       s0, s1, Nq, H, ldq, ldo, nqb, sl2, batch, blk);
  return (int)hipGetLastError();
}

CE_API int ce_attention_2seg_vt_bf16(const void* Q, const void* K1, const void* V1t, int len1, int ldk1, int ldv1t, int vt_cols1,
                                         const void* K2, const void* V2t, int len2, int ldk2, int ldv2t, int vt_cols2, void* O, int Nq,
                                         int H, int head_dim, int ldq, int ldo, float softmax_scale, int batch, hipStream_t stream) {
  if (!O) return CE_ERR_ARG;
  return attention_2seg_vt_launch(Q, K1, V1t, len1, ldk1, ldv1t, vt_cols1, K2, V2t, len2, ldk2, ldv2t, vt_cols2, O, Nq, H, head_dim, ldq, ldo,
                                  softmax_scale, batch, nullptr, nullptr, 0, stream);
}

/* The same attention with the output written as the MX fp8 operand of the out-projection that follows it in the fp8 mode: o8 e4m3
 * [batch Nq][ldo8] + E8M0 scales per 32 channels in the tiled layout of ce_gemm_mxfp8 (rows = batch Nq, K = H head_dim) - bit-identical to
 * ce_attention_2seg_vt_bf16 followed by ce_quant_rows_mxfp8. */
CE_API int ce_attention_2seg_vt_quant_bf16(const void* Q, const void* K1, const void* V1t, int len1, int ldk1, int ldv1t, int vt_cols1,
                                               const void* K2, const void* V2t, int len2, int ldk2, int ldv2t, int vt_cols2, void* o8,
                                               void* scale8, int Nq, int H, int head_dim, int ldq, int ldo8, float softmax_scale, int batch,
                                               hipStream_t stream) {
  if (!o8 || !scale8 || (ldo8 & 15) || ((H * head_dim) & 127)) return CE_ERR_ARG;
  return attention_2seg_vt_launch(Q, K1, V1t, len1, ldk1, ldv1t, vt_cols1, K2, V2t, len2, ldk2, ldv2t, vt_cols2, nullptr, Nq, H, head_dim, ldq,
                                  8, softmax_scale, batch, o8, scale8, ldo8, stream);
}
