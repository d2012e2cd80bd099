/* C ABI of libchronoedit_hip.so — the MI355X (gfx950) kernels behind the ChronoEdit DiT hot path.
 *
 * Conventions (SURVEY.md §8b "C-ABI for the HIP library"):
 *   - plain pointers + sizes; no torch types; every pointer is a DEVICE pointer unless noted;
 *   - no allocation, no synchronisation, no host reads inside: every launcher only enqueues
 *     kernels on `stream`, so whole denoising steps are hipGraph-capturable;
 *   - the caller owns every buffer; the library is stateless;
 *   - return value: 0 = ok, CE_ERR_* (<0) = rejected arguments (nothing was launched),
 *     >0 = hipError_t of the launch.  Never throws.
 *   - bf16 tensors are raw 16-bit patterns (`void*`); "ld*" are row strides in ELEMENTS.
 *
 * Each entry point cites the reference interface (file:line under /root/reference) it replaces.
 */
#ifndef CHRONOEDIT_HIP_H
#define CHRONOEDIT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef __HIP_PLATFORM_AMD__
typedef struct ihipStream_t* hipStream_t;
#endif

#define CE_OK 0
#define CE_ERR_ARG (-1)
#define CE_ERR_SHAPE (-2)
#define CE_ERR_ALIGN (-3)

/* GEMM epilogues */
#define CE_EPI_BIAS 0      /* C = bf16(A.W^T + bias) */
#define CE_EPI_BIAS_GELU 1 /* C = bf16(gelu_tanh(bf16(A.W^T + bias)))   diffusers FeedForward("gelu-approximate") */
#define CE_EPI_GATE_RES 2  /* C = bf16(res + bf16(A.W^T + bias) * gate[n]); gate == NULL -> 1 */
#define CE_EPI_BIAS_GELU_ERF 3 /* exact-erf GELU: diffusers FeedForward("gelu") of the image embedder */
#define CE_EPI_BIAS_ROW 6      /* C = bf16(A.W^T + bias[m]): bias along the ROWS of C - a product taken with the operand roles swapped
                                * (V^T = W_v.X^T: the attention kernel's V^T operand straight out of the projection, no transpose pass) */
#define CE_EPI_BIAS_T 7        /* C is [N][ldc] and holds the TRANSPOSE: C[n][m] = bf16(A.W^T[m][n] + bias[n]) - the same V^T as CE_EPI_BIAS_ROW
                                * gives with the operands swapped, but M stays the token count (round 6: 760 tiles of 384 rows = 2.97 rounds of
                                * 256 CUs where the swapped product is 1140 tiles of 256 = 4.45 rounds with a split-K tail); shapes the large-tile
                                * kernel does not take run as that swapped product: the same sums either way.  M % 8 == 0, N % 8 == 0 */
#define CE_EPI_F32 4           /* C is float* (ldc in floats): raw fp32 A.W^T, no bias (VAE mid-block attention scores) */

/* y = LayerNorm_fp32(x, eps) * a[d] + b[d] -> bf16.   One wave64 per row; D % 8 == 0, D <= 5120.
 * ab_rows > 0: row m uses a/b + (m / ab_rows) * ab_stride (one AdaLN vector pair per sample when samples are stacked).
 * Replaces `(self.norm1(h.float()) * (1 + scale) + shift).type_as(h)` and FP32LayerNorm(affine)
 * (chronoedit_diffusers/transformer_chronoedit.py:279,284,289,460). */
int ce_ln_affine_bf16(const void* x, void* y, const float* a, const float* b, int M, int D, int ldx, int ldy, float eps,
                      int ab_rows, int ab_stride, hipStream_t stream);

/* In place: x = RMSNorm_across_heads(x; w, eps), then (cos_sin != NULL) 3-D RoPE on (even, odd)
 * channel pairs of every head; cos_sin = [rope_rows or M][head_dim/2][2] fp32 (cos, sin), row m uses entry m % rope_rows.
 * (x2, w2) optional: a second tensor with the same geometry handled by the same launch (q and k of a fused buffer).
 * Replaces attn.norm_q / norm_k / norm_added_k + apply_rotary_emb
 * (transformer_chronoedit.py:62-65,73-79,85). */
int ce_rmsnorm_rope_bf16(void* x, const float* w, void* x2, const float* w2, const float* cos_sin, int M, int D, int ld,
                         int head_dim, float eps, int rope_rows, hipStream_t stream);

/* C[M,N] = epilogue(A[M,K] . W[N,K]^T + bias[N]); bf16 in/out, fp32 accumulate on MFMA.
 * K % 64 == 0, N % 8 == 0; res may alias C.  gate_rows > 0: row m uses gate[(m / gate_rows) * N + n] (one gate
 * vector per sample when several samples' tokens are stacked along M); gate_rows == 0: one gate vector.  Replaces every nn.Linear on the path
 * (transformer_chronoedit.py:58-60,84-86,106, FeedForward :262, patch_embedding :429, proj_out :461). */
int ce_gemm_bf16(const void* A, const void* W, void* C, const float* bias, int epilogue, const float* gate, const void* res,
                 int M, int N, int K, int lda, int ldw, int ldc, int ldres, int gate_rows, hipStream_t stream);

/* ce_gemm_bf16 with a K-SEGMENTED A operand: column k of A lives at A + (k / a_seg_k) * a_seg_stride + m * lda + k % a_seg_k
 * (elements; a_seg_k % 64 == 0, K % a_seg_k == 0; a_seg_k == 0 is ce_gemm_bf16).  This is the layout the second Ulysses
 * all-to-all leaves the attention output in - [source rank = head group][local token][D / W] - so the out-projection reads
 * it in place instead of a gather pass (reference design: chronoedit_diffsynth/wan_video_new_chronoedit.py:330-355; the
 * reference's xfuser path materialises the gathered tensor). */
int ce_gemm_aseg_bf16(const void* A, const void* W, void* C, const float* bias, int epilogue, const float* gate, const void* res,
                      int M, int N, int K, int lda, int ldw, int ldc, int ldres, int gate_rows, int a_seg_k,
                      long long a_seg_stride, hipStream_t stream);

/* ... with BOTH operands K-segmented (w_seg_k, w_seg_stride as for A).  seg_k = 64 and ld = 64 is the K-slab-major packing
 * [K/64][rows][64]: every 16 KiB half-tile the 256-tile kernel streams by LDS-DMA is then ONE contiguous block (measured
 * L2->LDS stream rate 21.7 vs 18.3 TB/s for the row-strided form, profiles/r02_l2_pattern_probe.txt).  Segmented W needs the
 * 256-tile kernel (K % 128 == 0), else CE_ERR_SHAPE. */
int ce_gemm_seg_bf16(const void* A, const void* W, void* C, const float* bias, int epilogue, const float* gate, const void* res,
                     int M, int N, int K, int lda, int ldw, int ldc, int ldres, int gate_rows, int a_seg_k, long long a_seg_stride,
                     int w_seg_k, long long w_seg_stride, hipStream_t stream);

/* ---- The alternative kernel bodies (A/B partners of the measurement tools and the body-equivalence tests) are NOT selectable through this
 * library: libchronoedit_hip.so picks every kernel from the call's own shape and keeps no state a launch reads except the caller-registered
 * split-K scratch below.  The selectors live in a second build of the same sources, libchronoedit_hip_diag.so
 * (include/chronoedit_hip_diag.h, -DCE_DIAGNOSTICS), which tools/ and tests/ load beside this one. ---- */


/* Which macro tile ce_gemm_bf16 runs a LARGE product on when the choice is automatic: 384 or 288 (x 256, ce_gemm384.hip) or 256 (x 256,
 * ce_gemm256w4.hip) rows - the one whose tile count falls better on `cus` compute units (full rounds + the last round, which is cut
 * along K when the split-K workspace of `ws_bytes` bytes allows it).  A pure function: no device is touched.  0: invalid arguments. */
int ce_gemm_bf16_tile_rows(int M, int N, int K, int cus, long long ws_bytes);

/* Scratch for the split-K tail of ce_gemm_bf16's 256-tile kernel: the tiles that would run as a partially filled last
 * round of workgroups are cut along K into fp32 slabs (256 KiB each, at most one per CU) and summed by a second
 * launch that applies the epilogue.  ptr is device memory owned by the caller (NULL switches the split off; that is
 * the default); the library never allocates.  Registers the DEFAULT scratch of the CURRENT device (one per device: two devices
 * in one process never share slabs).  Host-side knob, returns CE_OK. */
int ce_set_gemm_workspace(void* ptr, size_t bytes);

/* The same for ONE stream of the current device: a launch on `stream` uses this scratch instead of the device default, so GEMMs
 * running concurrently on several streams of a device do not meet in one buffer (the library keeps no other state a launch writes).
 * ptr == NULL unregisters (the stream falls back to the device default).  At most 32 registrations are kept per process; a 33rd
 * replaces the oldest.  SURVEY section 8(b): "workspace passed by caller, library stateless" - the registry is caller-owned memory
 * keyed by the caller's own stream handles.  Returns CE_OK. */
int ce_set_gemm_workspace_stream(hipStream_t stream, void* ptr, size_t bytes);

/* O = softmax(Q K1^T * scale) V1 [ + softmax(Q K2^T * scale) V2 ], per head, head_dim == 128, bf16.
 * Each segment's result is rounded to bf16 before the add (SDPA output dtype).
 * Replaces F.scaled_dot_product_attention at transformer_chronoedit.py:91-104. */
int ce_attention_bf16(const void* Q, const void* K1, const void* V1, int len1, int ldk1, int ldv1, const void* K2,
                      const void* V2, int len2, int ldk2, int ldv2, void* O, int Nq, int H, int head_dim, int ldq, int ldo,
                      float softmax_scale, hipStream_t stream);

/* The same for `batch` samples stacked along the rows of every operand (sample b owns rows [b Nq, (b+1) Nq) of Q / O
 * and rows [b len, (b+1) len) of each K/V segment; strides as above), in ONE launch: the cond / uncond forwards of a
 * classifier-free-guidance step (pipeline_chronoedit.py:715-735) are independent, and 2x the workgroups per launch
 * halves the partially filled last round.  batch == 1 is ce_attention_bf16. */
int ce_attention_batched_bf16(const void* Q, const void* K1, const void* V1, int len1, int ldk1, int ldv1, const void* K2,
                              const void* V2, int len2, int ldk2, int ldv2, void* O, int Nq, int H, int head_dim, int ldq,
                              int ldo, float softmax_scale, int batch, hipStream_t stream);

/* The same self-attention (one KV segment, `batch` samples per launch) with V handed over TRANSPOSED: Vt [H * 128][ldvt] bf16, row =
 * head channel, column = key, sample b's keys in columns [b len, (b + 1) len); ldvt >= (batch - 1) len + 64 ceil(len / 64), every
 * column up to ldvt finite (ce_v_transpose_bf16 zeroes the padding); len even when batch > 1 (dword-aligned sample offsets).
 * K and V^T tiles both reach LDS by LDS-DMA - no register
 * staging, no in-kernel transpose; the arithmetic and its order per row are those of ce_attention_batched_bf16.
 * Replaces the same F.scaled_dot_product_attention call (transformer_chronoedit.py:91-96). */
int ce_attention_vt_bf16(const void* Q, const void* K, const void* Vt, int len, int ldk, int ldvt, void* O, int Nq, int H, int head_dim,
                         int ldq, int ldo, float softmax_scale, int batch, hipStream_t stream);

/* Cross-attention - TWO key / value segments with a softmax each, the two outputs added in bf16 (transformer_chronoedit.py:91-104: text
 * and image keys) - with both V operands handed over transposed: V1t [H*128][ldv1t], V2t [H*128][ldv2t], sample b's keys at columns
 * [b*vt_cols, b*vt_cols + len) of its segment's V^T (vt_cols even, >= len; ldv*t >= (batch-1)*vt_cols + 64*ceil(len/64); every column
 * read - a sample's tail strip runs into the next sample's keys or the padding - is finite and meets P = 0).  K1 / K2 as in
 * ce_attention_batched_bf16 (samples stacked along the rows).  The K and V^T tiles of both segments reach LDS by LDS-DMA. */
int ce_attention_2seg_vt_bf16(const void* Q, const void* K1, const void* V1t, int len1, int ldk1, int ldv1t, int vt_cols1, const void* K2,
                              const void* V2t, int len2, int ldk2, int ldv2t, int vt_cols2, void* O, int Nq, int H, int head_dim, int ldq,
                              int ldo, float softmax_scale, int batch, hipStream_t stream);

/* ce_attention_2seg_vt_bf16 with its output written as the MX fp8 operand of the out-projection that follows it in the fp8 mode
 * (transformer_chronoedit.py:106 `attn.to_out[0]`; no counterpart in the reference): o8 e4m3 [batch Nq][ldo8] (ldo8 % 16 == 0) + one
 * E8M0 scale per 32 channels in the tiled layout of ce_gemm_mxfp8 (rows = batch Nq, K = H head_dim, K % 128 == 0) - bit-identical to
 * ce_attention_2seg_vt_bf16 followed by ce_quant_rows_mxfp8 (a 32-channel block of a query row is two lanes' accumulator registers). */
int ce_attention_2seg_vt_quant_bf16(const void* Q, const void* K1, const void* V1t, int len1, int ldk1, int ldv1t, int vt_cols1, const void* K2,
                                    const void* V2t, int len2, int ldk2, int ldv2t, int vt_cols2, void* o8, void* scale8, int Nq, int H,
                                    int head_dim, int ldq, int ldo8, float softmax_scale, int batch, hipStream_t stream);

/* ce_attention_vt_bf16 over the BLOCKED row layout an all-to-all leaves behind when every rank sent [sample][local token] rows
 * (Ulysses sequence parallelism with the guidance pair batched: chronoedit_amd/parallel.py): token g of sample b sits in row
 * (g / blk_rows) blk_stride + b blk_rows + g % blk_rows of Q, K and O (blk_rows = tokens per rank, a multiple of 64; blk_stride =
 * batch * blk_rows); Nq = query tokens per sample (a multiple of blk_rows), len = VALID keys per sample (the tail of the last block
 * is padding); Vt plain per sample, sample b's keys at columns [b vt_sample_cols, ...) (ce_v_transpose_blocked_bf16).  A key tile
 * never straddles a block, so the layout costs one scalar multiply-high per tile.  Replaces the gather + permute copies around
 * F.scaled_dot_product_attention in the reference's sequence-parallel path (wan_video_new_chronoedit.py:330-355 via xfuser). */
int ce_attention_vt_blocked_bf16(const void* Q, const void* K, const void* Vt, int len, int ldk, int ldvt, void* O, int Nq, int H, int head_dim,
                                 int ldq, int ldo, float softmax_scale, int batch, int blk_rows, int blk_stride, int vt_sample_cols,
                                 hipStream_t stream);

/* v in the blocked row layout above -> vt [H * 128][ldvt] plain per sample (sample b at columns [b vt_sample_cols, ...), columns past
 * n_keys zeroed); vt_sample_cols a multiple of 64, ldvt >= batch * vt_sample_cols. */
int ce_v_transpose_blocked_bf16(const void* v, int ldv, void* vt, int ldvt, int n_keys, int H, int batch, int blk_rows, int blk_stride,
                                int vt_sample_cols, hipStream_t stream);

/* v [n_keys][ldv] bf16 (head h at columns [128 h, 128 h + 128)) -> vt [H * 128][ldvt] (ldvt >= n_keys, multiple of 8; columns
 * [n_keys, ldvt) zeroed): the producer of ce_attention_vt_bf16's V operand. */
int ce_v_transpose_bf16(const void* v, int ldv, void* vt, int ldvt, int n_keys, int H, hipStream_t stream);


/* out[dim] = [cos(t f_i), sin(t f_i)], f_i = 1e4^(-i/(dim/2)), fp32; t is a device int64.
 * Replaces diffusers Timesteps(flip_sin_to_cos=True, downscale_freq_shift=0)
 * (transformer_chronoedit.py:137,153). */
int ce_timestep_sinusoid(const int64_t* t, float* out, int dim, hipStream_t stream);

/* Same table from a FLOATING-POINT timestep (device fp32): the reference's two sibling stacks hand the scheduler's float
 * timesteps to sinusoidal_embedding_1d (wan_video_dit_chronoedit.py:83-87, wan2pt1.py:191-200; call sites
 * wan_video_new_chronoedit.py:1383, wan2pt1.py:812). */
int ce_timestep_sinusoid_f32(const float* t, float* out, int dim, hipStream_t stream);

/* y[N] = post(W[N,K] . pre(x[K]) + bias); W fp32 (w_is_bf16 = 0) or bf16.
 * flags: 1 = pre: x <- bf16(silu(x)); 2 = post: silu; 4 = post: round result to bf16 (still stored fp32).
 * Replaces TimestepEmbedding + act_fn + time_proj (transformer_chronoedit.py:153-159). */
int ce_gemv(const void* W, int w_is_bf16, const float* x, const float* bias, float* y, int N, int K, int flags,
            hipStream_t stream);

/* mod[L][J][D] = table[L][J][D] + v[(v_rows==1 ? 0 : j)][D], plus 1.0 on rows j with bit j of one_mask set.
 * Replaces `(scale_shift_table + temb.float()).chunk(6)` and the `(1 + scale)` terms
 * (transformer_chronoedit.py:274-279,289,451,460). */
int ce_modulation(const float* table, const float* v, float* mod, int L, int J, int D, int v_rows, int one_mask,
                  hipStream_t stream);

/* im2col of the k = s = (1,2,2) patch Conv3d: x [C][T][H][W] -> cols [T*(H/2)*(W/2)][Kpad] (zero padded).
 * Replaces patch_embedding + flatten/transpose (transformer_chronoedit.py:429-430). */
int ce_patchify_bf16(const void* x, void* cols, int C, int T, int H, int W, int Kpad, hipStream_t stream);

/* The same for token rows [row0, row0 + nrows) only (cols has nrows rows; rows past the last token are zero): the local
 * shard of a sequence-parallel rank (zero pad: wan_video_new_chronoedit.py:1450-1453). */
int ce_patchify_rows_bf16(const void* x, void* cols, int C, int T, int H, int W, int Kpad, int row0, int nrows, hipStream_t stream);

/* Ulysses send side fused into the RMSNorm(+RoPE) pass.  For each of nt (<= 3) column blocks of x [M][ldx] (block i = columns
 * [col_i, col_i + D)): w_i != NULL -> RMSNorm across the D channels * w_i, then RoPE when cos_sin != NULL (as
 * ce_rmsnorm_rope_bf16); w_i == NULL -> copy (v).  Output: send[r][m][i][D/W] for destination rank r = n / (D/W), i.e. the
 * contiguous per-destination chunks all_to_all_single needs (no permute pass).  Replaces the q/k/v head-scatter of the
 * reference's sequence-parallel attention (wan_video_new_chronoedit.py:330-355 via xfuser) after
 * transformer_chronoedit.py:62-79. */
int ce_rope_scatter_bf16(const void* x, int ldx, void* send, int M, int D, int W, int nt, int col0, const float* w0, int col1,
                         const float* w1, int col2, const float* w2, const float* cos_sin, int head_dim, float eps,
                         int rope_rows, hipStream_t stream);

/* y [N][ldy] (col = (dh*2+dw)*Cout + c) -> out [Cout][T][H][W].
 * Replaces the reshape/permute/flatten at transformer_chronoedit.py:463-467. */
int ce_unpatchify_bf16(const void* y, void* out, int Cout, int T, int H, int W, int ldy, hipStream_t stream);

/* One denoising-loop tail, fused over the latents (n elements):
 *   v   = v_uncond ? bf16(u + bf16(g * bf16(c - u))) : c            pipeline_chronoedit.py:736
 *   x0  = x - sigma * v                                              fm_solvers_unipc.py:335-337
 *   xc  = use_corr ? a0*x_last + a1*m0 + a2*m1 + a3*x0 : x           UniC, fm_solvers_unipc.py:501-641
 *   x   = p0*xc + p1*x0 + p2*m0 ; x_last = xc ; m1 = m0 ; m0 = x0    UniP + history, :365-499,706-751
 * coef = device float[10] {g, sigma, use_corr, a0..a3, p0..p2} precomputed on the host per step.
 * flags bit0: round sigma*v to bf16 (the reference multiplies a 0-dim fp32 sigma into a bf16 tensor).
 * flags bit1: round the stored x, x_last and m0 to bf16 values (the reference keeps latents and scheduler history as bf16
 *             tensors, pipeline_chronoedit.py:681,739): the "reference-precision trajectory" mode of the pipeline.
 * Replaces scheduler.step + the CFG line of ChronoEditPipeline.__call__ (pipeline_chronoedit.py:736-739). */
int ce_cfg_unipc_step(const void* v_cond, const void* v_uncond, float* x, float* x_last, float* m0, float* m1,
                      float* x0_out, const float* coef, const void* reserved, long long n, int flags, hipStream_t stream);

/* ---- Wan-2.1 VAE (chronoedit/_src/tokenizers/wan2pt1.py; call sites pipeline_chronoedit.py:442,776-781) ----------
 * Activation frames are channels-last with a 1-pixel zero border: [H+2][W+2][C] bf16. */

/* Implicit-GEMM conv: out[t][h][w][co] = bias[co] + sum_{kt,kh,kw,ci} W[co][(kt,kh,kw)][ci] *
 *   in_frames[t*st + kt][h*ss + kh + in_off_h][w*ss + kw + in_off_w][ci]   (+ res_frames[t][...] when given).
 * in_frames / out_frames / res_frames are HOST arrays of device frame pointers (<= 16): temporal causal padding is
 * expressed by listing cache / zero frames in front.  weight [Cout][KT*KH*KW][Cin] bf16; Cin % 32 == 0, Cout % 8 == 0.
 * Output pixel (h,w) is stored at [(h+out_border)*out_Wp + (w+out_border)]*out_cstride + out_coff + co.
 * Replaces CausalConv3d / nn.Conv2d of Encoder3d/Decoder3d (wan2pt1.py:42-60,99-112,195-203,237-238). */
int ce_conv_igemm_bf16(const void* const* in_frames, int n_in_frames, const void* weight, const float* bias,
                       void* const* out_frames, int n_out_frames, const void* const* res_frames, int Cin, int Cout, int KT,
                       int KH, int KW, int st, int ss, int H_out, int W_out, int in_Wp, int in_off_h, int in_off_w, int out_Wp,
                       int out_border, int out_cstride, int out_coff, hipStream_t stream);

/* The decoder's head conv (Decoder3d.head, wan2pt1.py:401-403: CausalConv3d(96, 3, 3, padding=1)) as a bandwidth kernel: stride-1
 * 3 x 3 x 3 (KT = 3) or 1 x 3 x 3 (KT = 1) of Cin = 96 onto Cout <= 4 channels.  Same frames, addressing and result as
 * ce_conv_igemm_bf16 with st = ss = 1, in_off = 0 and no residual (weight [>= Cout][KT*9][96] bf16, bias fp32 [>= Cout] or NULL);
 * channels Cout..3 of an output pixel - and 4..7 when out_cstride - out_coff >= 8 - are written as zeros.  The three kernel rows ride
 * in the matrix instruction's output rows, so one pass over ten input rows yields eight output rows. */
int ce_conv3d_head_bf16(const void* const* in_frames, int n_in_frames, const void* weight, const float* bias, void* const* out_frames,
                        int n_out_frames, int Cin, int Cout, int KT, int H_out, int W_out, int in_Wp, int out_Wp, int out_border,
                        int out_cstride, int out_coff, hipStream_t stream);

/* The stride-1 3 x 3 (KT = 1) / 3 x 3 x 3 (KT = 3) convolutions of wide layers on the 256 x 256 x 64 LDS-DMA GEMM (same result as
 * ce_conv_igemm_bf16 with st = ss = 1, in_off = 0, out_border = 1): on bordered frames of one contiguous stack the A rows of the
 * implicit GEMM are linear in memory and the taps are constant offsets, so the product runs on the large-tile kernel and the border
 * positions it computes along the way are zeroed again.
 *   in_stack  [T_out + KT - 1 frames + ONE zeroed slack frame][H+2][W+2][Cin] bf16: for KT = 3 the first two frames are the causal
 *             padding / cache frames (CausalConv3d, wan2pt1.py:42-60), frame t of the output reads frames t .. t + KT - 1
 *   weight    [Cout][ldw] bf16: column ((kt*3 + kh) * S + kw * Cin + ci), S = 3*Cin rounded up to a multiple of 64 (zero columns in
 *             between when Cin = 96), zero from KT*3*S up to ldw >= the K-tile count rounded up to even x 64
 *   out_stack [T_out][H+2][W+2][out_cstride] (channels [0, Cout) written, borders zeroed); res_stack: same geometry, added, or NULL
 *   n_tile    256, 128 or 96: width of the macro tile (256 x 256 with 128 x 128 wave tiles | 256 x 128 with 128 x 64 | 256 x 96 with
 *             128 x 48); 1 (round 6; Cout == 96 and Cin 32, 96 or 192 only): not a GEMM tile - 512 positions x 96 channels per workgroup with
 *             the input slab itself in the LDS, the kw taps as position offsets (the same weight matrix, the same result up to the
 *             summation order), two waves per SIMD; 2: the same with one wave per SIMD (its A/B partner); 0 = 1 where it applies (the
 *             full-resolution layers of the VAE), else whichever tile wastes less of Cout
 * Cin % 32 == 0, Cout % 8 == 0. */
int ce_conv3d_gemm_bf16(const void* in_stack, const void* weight, int ldw, const float* bias, void* out_stack, const void* res_stack,
                        int T_out, int H, int W, int Cin, int Cout, int KT, int out_cstride, int n_tile, hipStream_t stream);

/* ce_conv3d_gemm_bf16 (same in_stack / weight / out_stack / res_stack operands and geometry) with the NEXT layer's RMS_norm (+ SiLU) applied in
 * the epilogue: with y = bf16(conv(in) + bias) [then bf16(res + y)], normed_stack = [silu]( RMS_norm(y) * gamma ) - ce_rms_silu_bf16's formula on
 * the rounded values it would have read.  Cout == 96 and Cin 32 / 96 / 192 only (the kernel whose lanes hold all 96 channels of a position);
 * gamma fp32 [96]; normed_stack has out_stack's geometry (borders zeroed).
 *   out_stack == NULL (then res_stack == NULL): y is never written - CausalConv3d -> RMS_norm -> SiLU inside a ResidualBlock (wan2pt1.py:195-200);
 *   out_stack != NULL: y to out_stack AND its normalised form to normed_stack (the next block's shortcut and first norm, :186-220, or the
 *                      head's norm, :401-403): the pass that would have re-read y is gone. */
int ce_conv3d_gemm_rms_silu_bf16(const void* in_stack, const void* weight, int ldw, const float* bias, void* out_stack, const void* res_stack,
                                 void* normed_stack, int T_out, int H, int W, int Cin, int Cout, int KT, int out_cstride, const float* gamma,
                                 int apply_silu, hipStream_t stream);

/* y = [silu]( x / max(||x||_2, 1e-12) * sqrt(C) * gamma ) per pixel over C channels; x, y are stacks of npix/(H*W) frames
 * with in_border / out_border zero borders (the interiors are written, never the border).  C % 8 == 0, C <= 512.
 * Replaces RMS_norm (+ nn.SiLU) (wan2pt1.py:63-75,193-200). */
int ce_rms_silu_bf16(const void* x, void* y, const float* gamma, long long npix, int C, int H, int W, int in_border,
                     int out_border, int apply_silu, hipStream_t stream);

/* Zero the one-pixel border of T frames [H+2][W+2][ld] (channels [0, C)): for buffers that were not zero-filled and whose producer
 * writes interiors only (the border is the zero padding of the next convolution). */
int ce_zero_border_bf16(void* frames, int T, int H, int W, int C, int ld, hipStream_t stream);

/* nearest-exact 2x spatial upsample of T bordered frames [H+2][W+2][C] -> [2H+2][2W+2][C] (wan2pt1.py:78-83,99-104). */
int ce_upsample2x_bf16(const void* x, void* y, int T, int C, int H, int W, hipStream_t stream);

/* O [Nq][ldo] = softmax(Q K^T * softmax_scale) V for ONE attention head of dimension C (128 or 384) as a single flash-style kernel
 * (no [Nq, Nk] score matrix in memory): Q [Nq][ldq], K [Nk][ldk] bf16 rows; Vt [C][ldvt] = V transposed, keys contiguous, ldvt >=
 * 64 ceil(Nk / 64) with the padding columns finite (zero).  Replaces the scaled_dot_product_attention of the Wan VAE's mid-block
 * AttentionBlock (chronoedit/_src/tokenizers/wan2pt1.py:247-255: one head over the h*w positions of a frame).
 * ws / ws_bytes (round 6; may be NULL / 0): caller-owned scratch for a split of the KEY axis - when the 128-row query blocks alone leave most
 * CUs idle (113 blocks at the 14 400 positions of a 720p frame) up to four workgroups share a query block's keys, write un-normalised fp32
 * partial results (nsplit * Nq * (C + 2) * 4 bytes) and a second launch merges them; without a workspace one workgroup walks all keys. */
int ce_attention_1head_bf16(const void* Q, const void* K, const void* Vt, void* O, int Nq, int Nk, int C, int ldq, int ldk, int ldvt,
                            int ldo, float softmax_scale, void* ws, long long ws_bytes, hipStream_t stream);

/* probs[m][0:npad] = bf16(softmax(scale * scores[m][0:n])) (zeros beyond n); scores fp32.  Mid-block attention
 * (wan2pt1.py:247-252) runs as GEMM (CE_EPI_F32) -> this -> GEMM. */
int ce_softmax_rows_f32_bf16(const float* scores, void* probs, int M, int n, int npad, int ld, int ldp, float scale,
                             hipStream_t stream);

/* ---- fp8 (OCP e4m3) path of BASELINE.json configs[4].  The reference has no fp8 inference code: the contract is defined
 * here (per-row activation scales, per-output-channel weight scales, fp32 accumulation on the MX matrix instruction) ---- */

/* q[m][k] = fp8_e4m3(x[m][k] / s[m]), s[m] = max_k |x[m][k]| / 448 (1 for an all-zero row); x bf16 [M][ldx], q bytes [M][ldq].
 * Used for activations (per token row) and, once at load time, for nn.Linear weights (per output channel). */
int ce_quant_rows_fp8(const void* x, void* q, float* scale, int M, int K, int ldx, int ldq, hipStream_t stream);

/* ce_ln_affine_bf16 followed by ce_quant_rows_fp8 in one pass: q / scale are the quantisation of the bf16 row that
 * ce_ln_affine_bf16 would have written (bit-identical to the two-launch form). */
int ce_ln_affine_fp8(const void* x, void* q, float* scale, const float* a, const float* b, int M, int D, int ldx, int ldq, float eps,
                     int ab_rows, int ab_stride, hipStream_t stream);

/* C = epilogue(sa[m] * sw[n] * (Aq Wq^T)[m][n] + bias[n]); Aq [M][lda], Wq [N][ldw] fp8 e4m3 bytes, C bf16; K % 256 == 0.
 * Epilogues 0 (bias), 1 (bias + tanh GELU), 2 (gated residual) as ce_gemm_bf16. */
int ce_gemm_fp8(const void* Aq, const void* Wq, void* C, const float* sa, const float* sw, const float* bias, int epilogue,
                const float* gate, const void* res, int M, int N, int K, int lda, int ldw, int ldc, int ldres, int gate_rows,
                hipStream_t stream);

/* ---- the MX form of the fp8 GEMMs (round 4; BASELINE.json configs[4] "fp8 weights"): OCP MXFP8 operands - e4m3 elements with one E8M0
 * scale per 32 consecutive K elements of a row, scale = the smallest power of two with amax / scale <= 448 (2^-126 for an all-zero block:
 * the non-saturating choice - with the OCP floor rule 2^(floor(log2 amax) - 8) an eighth of the blocks have their largest elements clipped),
 * elements RNE(x / scale) - and the block scales applied INSIDE the matrix pipe by v_mfma_scale_f32_16x16x128_f8f6f4.
 * Scale bytes are stored in the order the GEMM reads them.
 * A operand (activations; every producer of this library writes it): [ceil(rows / 128)][K / 128][4][16][8], i.e. the scale of elements
 * [128 t + 32 g, + 32) of row r at byte ((r / 128 * (K / 128) + t) * 4 + g) * 128 + (r % 16) * 8 + (r / 16) % 8.
 * W operand (weights; round 6): [ceil(rows / 128)][K / 128][4][128], the same byte with r % 128 in place of (r % 16) * 8 + (r / 16) % 8 - the
 * GEMM feeds fragment G of a wave tile with the weight rows G + 8 i, so that a lane's eight accumulators are eight CONSECUTIVE output columns
 * (the epilogue stores straight from registers, csrc/ce_gemm_fp8w4.hip), and a lane's eight scale bytes are consecutive in this order.
 * A scale buffer holds ceil(rows / 128) * (K / 128) * 512 bytes in either order.
 * The reference has no fp8 path: the contract is oracle.dit_oracle.mx_quant / linear_mxfp8. ---- */

/* x bf16 [M][ldx] -> q e4m3 bytes [M][ldq] + scale8 in the A order (above).  K % 128 == 0.  Activations (per token row). */
int ce_quant_rows_mxfp8(const void* x, void* q, void* scale8, int M, int K, int ldx, int ldq, hipStream_t stream);

/* The same with scale8 in the W order: the weight operand of ce_gemm_mxfp8 / ce_gemm_mxfp8_gelu_quant, quantised once at load time. */
int ce_quant_rows_mxfp8_w(const void* x, void* q, void* scale8, int M, int K, int ldx, int ldq, hipStream_t stream);

/* ce_ln_affine_bf16 followed by ce_quant_rows_mxfp8 in one pass (the quantisation of the bf16 row ce_ln_affine_bf16 would have written). */
int ce_ln_affine_mxfp8(const void* x, void* q, void* scale8, const float* a, const float* b, int M, int D, int ldx, int ldq, float eps,
                       int ab_rows, int ab_stride, hipStream_t stream);

/* C = epilogue(sum_blocks 2^(ea + ew) (Aq Wq^T)_block + bias[n]); Aq [M][lda], Wq [N][ldw] e4m3 bytes with their scale buffers sa8 (A order) / sw8 (W order),
 * C bf16; K % 256 == 0, N % 8 == 0.  Epilogues 0 (bias), 1 (bias + tanh GELU), 2 (gated residual; gate rows per sample >= 256 or one gate)
 * as ce_gemm_bf16; the one-wave-per-SIMD main loop of csrc/ce_gemm_fp8w4.hip, split-K tail through the ce_set_gemm_workspace scratch. */
int ce_gemm_mxfp8(const void* Aq, const void* Wq, void* C, const void* sa8, const void* sw8, const float* bias, int epilogue, const float* gate,
                  const void* res, int M, int N, int K, int lda, int ldw, int ldc, int ldres, int gate_rows, hipStream_t stream);

/* The FFN-up form: q_out / qs_out = ce_quant_rows_mxfp8( bf16( gelu_tanh( bf16( Aq Wq^T + bias ) ) ) ) - the bias + GELU epilogue emits the
 * NEXT GEMM's MX operand directly (bit-identical to ce_gemm_mxfp8 with epilogue 1 followed by ce_quant_rows_mxfp8; the bf16 matrix is
 * never written: an MX block is 32 consecutive output columns = four adjacent 16-byte chunks of the staged row, so its scale needs no
 * row-wide reduction).  q_out e4m3 bytes [M][ldq], qs_out the tiled scales of an [M][N] operand.  N % 128 == 0, K % 256 == 0. */
int ce_gemm_mxfp8_gelu_quant(const void* Aq, const void* Wq, const void* sa8, const void* sw8, const float* bias, void* q_out, void* qs_out,
                             int M, int N, int K, int lda, int ldw, int ldq, hipStream_t stream);


/* ---- MXFP8 self-attention of the fp8 mode ("fp8 weights+attn", BASELINE.json configs[4]).  Contract (csrc/ce_attn_fp8.hip,
 * oracle/dit_oracle.py::attention_mxfp8): Q, K quantised to OCP MXFP8 - e4m3 elements, one E8M0 scale per 32 head channels - V per
 * 32 keys, products on v_mfma_scale_f32_32x32x64_f8f6f4, P = exp2(S - offset) <= 8 in e4m3 (offset = a row maximum - 3), fp32
 * accumulation; replaces
 * F.scaled_dot_product_attention at transformer_chronoedit.py:91-104 in that mode. ---- */

/* ce_rmsnorm_rope_bf16 that writes MXFP8 instead of bf16: q8 [M][ldq8] e4m3 bytes, scale8 [M][D/32] E8M0 bytes (x is not modified);
 * the bf16-rounded value is multiplied by post_scale before the quantisation - softmax_scale * log2(e) for q (the scores then leave
 * the matrix pipe in the exp2 domain), 1 for k.  transformer_chronoedit.py:62-65,73-79. */
int ce_rmsnorm_rope_mxfp8(const void* x, const float* w, const float* cos_sin, void* q8, void* scale8, int M, int D, int ldx, int ldq8,
                          int head_dim, float eps, int rope_rows, float post_scale, hipStream_t stream);

/* V [batch * n_tokens][ldv] bf16 (head h = columns 128 h ..) -> V^T tiles v8t [batch][H][128][npad] e4m3 bytes + sv
 * [batch][H][npad/64][128][2] E8M0 bytes (npad = n_tokens rounded up to 64, zero keys at the end).  Inside every 64-key tile position
 * 32 g + j holds key 32 (j >> 4) + (j & 3) + 8 ((j & 15) >> 2) + 4 g - the order of the P operand in the matrix unit's accumulator
 * registers; one scale per 32 CONSECUTIVE keys (sv[..][t][d][beta] covers keys 64 t + 32 beta .. of channel d). */
int ce_v_mxfp8_transpose(const void* v, int ldv, void* v8t, void* sv, int n_tokens, int batch, int H, int npad, hipStream_t stream);

/* O [batch * Nq][ldo] bf16 = attention over the operands above (q8 / k8 row strides in bytes, head h at byte column 128 h; sq / sk
 * [rows][H * 4]); head_dim == 128. */
int ce_attention_mxfp8(const void* q8, const void* sq, const void* k8, const void* sk, const void* v8t, const void* sv, void* O, int Nq,
                       int Nkv, int npad, int H, int head_dim, int ldq8, int ldk8, int ldo, int batch, hipStream_t stream);

/* ce_attention_mxfp8 with its output written as the out-projection's MX fp8 operand (o8 / scale8 as in ce_attention_2seg_vt_quant_bf16):
 * bit-identical to ce_attention_mxfp8 followed by ce_quant_rows_mxfp8; always the software-pipelined body. */
int ce_attention_mxfp8_quant(const void* q8, const void* sq, const void* k8, const void* sk, const void* v8t, const void* sv, void* o8,
                             void* scale8, int Nq, int Nkv, int npad, int H, int head_dim, int ldq8, int ldk8, int ldo8, int batch,
                             hipStream_t stream);

/* Second segment of a TWO-segment attention under the same contract - the cross-attention, out = bf16(SDPA(q, k_text, v_text)) +
 * bf16(SDPA(q, k_image, v_image)) (transformer_chronoedit.py:96-107): as ce_attention_mxfp8 / ce_attention_mxfp8_quant on the second
 * segment's operands, with the bf16 rows o_add [batch Nq][ldadd] (the first segment's result, from a plain ce_attention_mxfp8 call) added to
 * this segment's bf16-rounded result.  The sum goes to O (bf16 [batch Nq][ldo]; may be o_add itself) or, quantised exactly as
 * ce_quant_rows_mxfp8 would quantise it, to o8 / scale8 (the out-projection's MX operand) - exactly one of O and o8 is non-NULL. */
int ce_attention_mxfp8_add(const void* q8, const void* sq, const void* k8, const void* sk, const void* v8t, const void* sv, const void* o_add,
                           int ldadd, void* O, int ldo, void* o8, void* scale8, int ldo8, int Nq, int Nkv, int npad, int H, int head_dim,
                           int ldq8, int ldk8, int batch, hipStream_t stream);



/* ---- conditioning encoders (run once per edit, outside the loop: pipeline_chronoedit.py:205-254; the arithmetic is
 * transformers==4.57.1 CLIPVisionModel / UMT5EncoderModel, restated in oracle/clip_oracle.py, oracle/umt5_oracle.py) ---- */

/* batch0 x batch1 independent C = A W^T products with two-level element strides (head within sample); epilogue 0 (bias,
 * bf16 C) or 4 (fp32 C, no bias).  The per-head Q K^T and P V products of the encoders' attention. */
int ce_gemm_batched_bf16(const void* A, const void* W, void* C, const float* bias, int epilogue, int M, int N, int K, int lda,
                         int ldw, int ldc, int batch0, int batch1, long long sA0, long long sA1, long long sW0, long long sW1,
                         long long sC0, long long sC1, hipStream_t stream);

/* cols[(b gh + py) gw + px][c P P + y P + x] = img[b][c][py P + y][px P + x], zero padded to Kpad columns: the im2col of
 * CLIPVisionEmbeddings.patch_embedding (Conv2d, kernel = stride = P, no bias). */
int ce_im2col_patch2d_bf16(const void* img, void* cols, int B, int C, int H, int W, int P, int Kpad, hipStream_t stream);

/* out[i][:] = table[ids[i]][:] (ids int64, clamped to [0, vocab)): UMT5Stack.embed_tokens. */
int ce_gather_rows_bf16(const void* table, const long long* ids, void* out, int n, int D, int ldt, int ldo, int vocab,
                        hipStream_t stream);

/* y = bf16(bf16(x * rsqrt(mean(x^2) + eps)) * w), statistics in fp32, w bf16: UMT5LayerNorm.forward. */
int ce_rmsnorm_bf16(const void* x, void* y, const void* w, int M, int D, int ldx, int ldy, float eps, hipStream_t stream);

/* probs[b][h][q][:] = softmax_k(scores[b][h][q][k] + table[bucket_lut[k - q + Lq - 1]][h]) over keys k < valid_len[b]
 * (fp32 in, bf16 out, columns [Lk, ldp) written as zeros); table / bucket_lut may both be NULL (no bias), valid_len may be
 * NULL (no padding mask).  UMT5Attention: position bias + extended attention mask + softmax in fp32. */
int ce_softmax_t5_bf16(const float* scores, void* probs, int batch, int heads, int Lq, int Lk, int ld, int ldp,
                       const int* bucket_lut, const float* table, const int* valid_len, hipStream_t stream);

/* ---- a RCCL communicator owned by the library (csrc/ce_comm.hip): the exchanges of the sequence-parallel forward as C-ABI calls on the
 * caller's stream.  Replaces the torch.distributed collectives of the reference's sequence-parallel path (xfuser's Ulysses all-to-all behind
 * chronoedit_diffsynth/wan_video_new_chronoedit.py:330-355, the final all_gather :1495-1498) where the caller needs a step with
 * exchanges under hipGraph capture: these calls create no Work objects and nothing polls them.  RCCL is not linked - ce_comm_load()
 * dlopen()s the librccl.so the process already uses (the one torch ships).  All return CE_OK or CE_ERR_ARG (the RCCL error text goes to
 * stderr). */
int ce_comm_load(const char* librccl_path);
/* 128 bytes: ncclGetUniqueId, called by ONE rank; the caller distributes the bytes to the other ranks (any transport). */
int ce_comm_unique_id(void* id128);
/* ncclCommInitRank on the CURRENT device - collective over the `world` ranks that share id128; *comm_out is the handle of the other calls. */
int ce_comm_init(void** comm_out, const void* id128, int rank, int world);
/* ncclCommDestroy + free of the handle.  UNSAFE on the stack this library was validated on (RCCL 2.26.6 inside torch 2.10, next to
 * torch.distributed's own process group): the call blocks for good even on a one-rank communicator (tools/owned_comm_probe.py,
 * profiles/r04_owned_comm_probe.txt), so the Python side never calls it - it keeps ONE communicator per (group, device) for the life of
 * the process instead (parallel.OwnedComm.get).  Exported for hosts whose RCCL tears down cleanly. */
int ce_comm_destroy(void* comm);
/* send / recv: `world` chunks of bytes_per_peer bytes, chunk p to / from rank p (all_to_all_single with equal splits), as one grouped
 * batch of ncclSend / ncclRecv on `stream`.  Buffers must stay valid until the stream has passed the call. */
int ce_comm_all_to_all(void* comm, const void* send, void* recv, size_t bytes_per_peer, hipStream_t stream);
/* recv = every rank's bytes_per_rank-byte block, in rank order (ncclAllGather on `stream`). */
int ce_comm_all_gather(void* comm, const void* send, void* recv, size_t bytes_per_rank, hipStream_t stream);

/* How the loaded library was compiled - a constant, no device is touched: bit 0 = the diagnostic build (libchronoedit_hip_diag.so,
 * include/chronoedit_hip_diag.h), bit 1 = -DF8_A3, bits 4-7 = F8_DMA_SCHED, bits 8-15 = F8_ABLATE (non-zero: a timing-only build of the MX fp8
 * GEMM with one ingredient compiled out; its results are garbage and chronoedit_amd.hiplib.load refuses it).  The product build returns 0x10. */
int ce_build_info(void);

#ifdef __cplusplus
}
#endif
#endif /* CHRONOEDIT_HIP_H */
