/* Diagnostic selectors of libchronoedit_hip_diag.so - the SAME sources as libchronoedit_hip.so compiled with -DCE_DIAGNOSTICS.
 *
 * The product library (include/chronoedit_hip.h) exports none of these: there every selector below is a compile-time constant at its
 * default, so two engines in one process cannot change each other's kernels.  The diagnostic build exports the product ABI PLUS these
 * process-wide A/B switches of alternative kernel bodies that compute the same results; only tools/ (measurement) and tests/ (body-
 * equivalence) load it (chronoedit_amd.hiplib.load_diagnostics, chronoedit_amd.ops.set_*).  Every setter returns the previous value.
 */
#ifndef CHRONOEDIT_HIP_DIAG_H
#define CHRONOEDIT_HIP_DIAG_H

#include "chronoedit_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Kernel selection for ce_gemm_bf16 (returns the previous setting): -1 automatic (default: large shapes on the one-wave-per-SIMD
 * LDS-DMA kernels, macro tile per ce_gemm_bf16_tile_rows), 0 force the 128x128 register-staged kernel; wherever the shape allows the
 * large-tile kernels: 1 the 8-wave 256x256 main loop (csrc/ce_gemm256.hip), 2 the same staggered, 3 / 4 / 5 the one-wave-per-SIMD
 * 256x256 main loop (csrc/ce_gemm256w4.hip; A ring of 3 stages / 2 stages / 3 stages and one barrier per K-tile), 6 the 384x256 macro
 * tile (csrc/ce_gemm384.hip), 7 its 288x256 form.  Host-side test/bench knob. */
int ce_set_gemm_variant(int variant);

/* Loop body behind ce_attention_bf16 / ce_attention_batched_bf16 (returns the previous value; all are tested against the
 * same reference): 0 automatic (= 64); 4 / 8 the plain kernel with 4 / 8 waves per workgroup; 64 software-pipelined, K by
 * LDS-DMA, pre-scaled Q, speculative softmax with an exact fall-back route per tile (default); 128 / 129: as 64, but the V^T form
 * (ce_attention_vt_bf16, ce_attention_vt_blocked_bf16) runs its one-wave-per-SIMD body (4 waves x 64 query rows, Q and O^T in the
 * accumulator file; bit-identical results) with one workgroup per work item / with one persistent workgroup per CU.  Other values
 * are ignored.  Host-side tuning knob. */
int ce_set_attention_waves(int nwave);

/* Main loop of ce_gemm_fp8 (returns the previous setting): 0 = 8 waves / 4 phases per K-tile (csrc/ce_gemm_fp8.hip), 1 = one wave per
 * SIMD (csrc/ce_gemm_fp8w4.hip: 4 waves, 128 x 128 wave tiles, accumulators in AGPRs, one barrier per K-tile).  Same results bit for
 * bit (same products, same summation order per accumulator).  Host-side test / bench knob. */
int ce_set_gemm_fp8_variant(int variant);

/* Loop body of ce_attention_mxfp8 (returns the previous value): 0 plain (exact running maximum every tile), 1 software-pipelined
 * with a speculative integer offset, row sums on the matrix pipe and an exact repair route per tile (default).  Other values are
 * ignored.  Host-side tuning knob. */
int ce_set_attention_mxfp8_variant(int variant);

/* Workgroups of the persistent form of the software-pipelined MXFP8 kernel (returns the previous value): n > 0 (a multiple of 8) -
 * n workgroups walk the work order with stride n (default 512 = 2 x #CUs: +2 % at 7 200 keys, +0.3 ... 0.5 % above); 0 - one
 * workgroup per (head, query block, sample).  Host-side tuning knob. */
int ce_set_attention_mxfp8_persistent(int n);

/* How often the attention kernels' speculative softmax fell back to its exact route: hits[0] = (wave, key tile) pairs of the bf16
 * software-pipelined kernels since the last reset, hits[1] = of the MXFP8 kernel; reset != 0 zeroes both counters.  Synchronises the
 * device.  tools/attn_peaked.py: the kernels' rate on peaked score rows. */
int ce_diag_attention_exact_route_hits(unsigned long long* hits, int reset);

#ifdef __cplusplus
}
#endif
#endif /* CHRONOEDIT_HIP_DIAG_H */
