// K19: classifier-free-guidance combine + one flow-UniPC (order <= 2, bh2, predict-x0) update,
// fused into a single pass over the latents.  HBM-bound: algorithmic bytes per element =
// 2*2 (two bf16 predictions) + 4*4 (x, x_last, m0, m1 read) + 4*4 (written back).
//
// Reference semantics:
//   pipeline_chronoedit.py:736      noise = uncond + g * (cond - uncond)          (bf16 arithmetic)
//   fm_solvers_unipc.py:335-337     x0 = sample - sigma_t * model_output
//   fm_solvers_unipc.py:501-641     UniC corrector  (linear in last_sample, m0, m1, x0)
//   fm_solvers_unipc.py:365-499     UniP predictor  (linear in sample, m0, m1)
//   fm_solvers_unipc.py:706-751     history shift / last_sample bookkeeping
// All scalar coefficients (host-side lambda/h/phi/rho math of :420-468,:560-620) are precomputed
// per step by chronoedit_amd/scheduler.py into a device table, so the update needs no host
// sync and is hipGraph-capturable.
//
// coef[0] = guidance scale g        coef[1] = sigma_t
// coef[2] = use_corrector (0/1)     coef[3..6]  = corrector weights on (x_last, m0, m1, x0)
// coef[7..9] = predictor weights on (x_corrected, x0_new (= new m0), m0_old (= new m1))
#include "ce_common.h"

__global__ __launch_bounds__(256) void cfg_unipc_kernel(const bf16* __restrict__ v_cond, const bf16* __restrict__ v_uncond,
                                                        float* __restrict__ x, float* __restrict__ x_last,
                                                        float* __restrict__ m0, float* __restrict__ m1,
                                                        float* __restrict__ x0_out, const float* __restrict__ coef,
                                                        long long n, int flags) {
  const float g = coef[0], sigma = coef[1];
  const bool use_corr = coef[2] != 0.f;
  const float a0 = coef[3], a1 = coef[4], a2 = coef[5], a3 = coef[6];
  const float p0 = coef[7], p1 = coef[8], p2 = coef[9];
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float v = (float)v_cond[i];
    if (v_uncond != nullptr) {
      const float u = (float)v_uncond[i];
      // bf16 tensor arithmetic of the reference: each op rounds to bf16
      v = round_bf16(u + round_bf16(g * round_bf16(v - u)));
    }
    const float xs = x[i];
    float x0 = xs - ((flags & 1) ? round_bf16(sigma * v) : sigma * v);
    // flags & 2 = reference-precision trajectory: the reference's latents, x0 prediction and scheduler history are bf16 TENSORS
    // (pipeline_chronoedit.py:681 passes torch.bfloat16 to prepare_latents), so x0 is rounded before UniC consumes it and the
    // corrected sample before UniP does; storage stays fp32, the values are bf16
    if (flags & 2) x0 = round_bf16(x0);
    const float m0o = m0[i], m1o = m1[i];
    float xc = xs;
    if (use_corr) {
      xc = a0 * x_last[i] + a1 * m0o + a2 * m1o + a3 * x0;
    }
    if (flags & 2) xc = round_bf16(xc);
    float xn = p0 * xc + p1 * x0 + p2 * m0o;
    if (flags & 2) xn = round_bf16(xn);
    const float x0s = x0;
    x[i] = xn;
    x_last[i] = xc;
    m1[i] = m0o;
    m0[i] = x0s;
    if (x0_out != nullptr) x0_out[i] = x0;
  }
}

CE_API int ce_cfg_unipc_step(const void* v_cond, const void* v_uncond, float* x, float* x_last, float* m0, float* m1,
                                 float* x0_out, const float* coef, const void* reserved, long long n, int flags,
                                 hipStream_t stream) {
  (void)reserved;
  if (!v_cond || !x || !x_last || !m0 || !m1 || !coef || n <= 0) return CE_ERR_ARG;
  long long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(cfg_unipc_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const bf16*)v_cond,
                     (const bf16*)v_uncond, x, x_last, m0, m1, x0_out, coef, n, flags);
  return (int)hipGetLastError();
}
