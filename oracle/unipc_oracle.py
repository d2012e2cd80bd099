"""CPU oracle for the flow-matching UniPC multistep scheduler (TEST INFRASTRUCTURE).

Restates /root/reference/chronoedit/_src/models/fm_solvers_unipc.py (the in-repo sibling of the
un-vendored diffusers UniPCMultistepScheduler(use_flow_sigmas=True) the pipeline uses,
scripts/run_inference_diffusers.py:379-382):
  sigma grid            :196-225      convert_model_output :293-347
  UniP predictor (bh2)  :365-499      UniC corrector       :501-641      step :670-756
Pinned by tests/golden/unipc_*.pt, produced by running the reference file itself
(oracle/gen_golden_unipc.py).  Plain tensor ops in the order the reference applies them, so the
fp32 results agree bit-for-bit.
"""
from __future__ import annotations

import numpy as np
import torch


class UniPCOracle:
    def __init__(self, num_train_timesteps=1000, solver_order=2, shift=1.0, lower_order_final=True, disable_corrector=()):
        self.T = num_train_timesteps
        self.solver_order = solver_order
        self.lower_order_final = lower_order_final
        self.disable_corrector = list(disable_corrector)
        alphas = np.linspace(1, 1 / num_train_timesteps, num_train_timesteps)[::-1].copy()
        sigmas = torch.tensor(1.0 - alphas).to(torch.float32)
        sigmas = shift * sigmas / (1 + (shift - 1) * sigmas)  # :124-127
        self.sigma_min, self.sigma_max = sigmas[-1].item(), sigmas[0].item()
        self.init_shift = shift

    def set_timesteps(self, n, shift=None, grid="sibling"):
        if grid == "sibling":  # fm_solvers_unipc.py:196-199
            sig = np.linspace(self.sigma_max, self.sigma_min, n + 1).copy()[:-1]
        else:  # diffusers 0.35.2 use_flow_sigmas grid (recalled)
            sig = (1.0 - np.linspace(1, 1 / self.T, n + 1))[::-1].copy()[:-1]
        if shift is None:
            shift = self.init_shift
        sig = shift * sig / (1 + (shift - 1) * sig)
        self.timesteps = torch.tensor(sig * self.T).to(torch.int64)
        self.sigmas = torch.tensor(np.concatenate([sig, [0]]).astype(np.float32))
        self.model_outputs = [None] * self.solver_order
        self.lower_order_nums = 0
        self.last_sample = None
        self.step_index = 0
        self.this_order = None

    @staticmethod
    def _lam(sigma):
        return torch.log(1 - sigma) - torch.log(sigma)

    def _coeffs(self, sigma_t, sigma_s0, order, rk_sigmas):
        lam_t, lam_s0 = self._lam(sigma_t), self._lam(sigma_s0)
        h = lam_t - lam_s0
        rks = [(self._lam(s) - lam_s0) / h for s in rk_sigmas] + [1.0]
        rks = torch.tensor(rks)
        hh = -h
        h_phi_1 = torch.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1
        B_h = torch.expm1(hh)
        R, b, fact = [], [], 1
        for i in range(1, order + 1):
            R.append(torch.pow(rks, i - 1))
            b.append(h_phi_k * fact / B_h)
            fact *= i + 1
            h_phi_k = h_phi_k / hh - 1 / fact
        return rks, torch.stack(R), torch.tensor(b), h_phi_1, B_h

    def step(self, model_output, sample):
        i = self.step_index
        use_corr = i > 0 and (i - 1) not in self.disable_corrector and self.last_sample is not None
        x0 = sample - self.sigmas[i] * model_output  # :335-337
        if use_corr:  # UniC :501-641
            order = self.this_order
            m0 = self.model_outputs[-1]
            sigma_t, sigma_s0 = self.sigmas[i], self.sigmas[i - 1]
            rk_s = [self.sigmas[i - (k + 1)] for k in range(1, order)]
            rks, R, b, h_phi_1, B_h = self._coeffs(sigma_t, sigma_s0, order, rk_s)
            D1s = [(self.model_outputs[-(k + 1)] - m0) / rks[k - 1] for k in range(1, order)]
            rhos_c = torch.tensor([0.5]) if order == 1 else torch.linalg.solve(R, b)
            x_t_ = sigma_t / sigma_s0 * self.last_sample - (1 - sigma_t) * h_phi_1 * m0
            corr = sum(rhos_c[k] * D1s[k] for k in range(len(D1s))) if D1s else 0
            sample = x_t_ - (1 - sigma_t) * B_h * (corr + rhos_c[-1] * (x0 - m0))
        for k in range(self.solver_order - 1):
            self.model_outputs[k] = self.model_outputs[k + 1]
        self.model_outputs[-1] = x0
        n = len(self.timesteps)
        order = min(self.solver_order, n - i) if self.lower_order_final else self.solver_order
        self.this_order = min(order, self.lower_order_nums + 1)
        self.last_sample = sample
        # UniP :365-499
        order = self.this_order
        m0 = self.model_outputs[-1]
        sigma_t, sigma_s0 = self.sigmas[i + 1], self.sigmas[i]
        rk_s = [self.sigmas[i - k] for k in range(1, order)]
        rks, R, b, h_phi_1, B_h = self._coeffs(sigma_t, sigma_s0, order, rk_s)
        D1s = [(self.model_outputs[-(k + 1)] - m0) / rks[k - 1] for k in range(1, order)]
        x_t_ = sigma_t / sigma_s0 * sample - (1 - sigma_t) * h_phi_1 * m0
        if D1s:
            assert order == 2
            pred = 0.5 * D1s[0]
        else:
            pred = 0
        prev = x_t_ - (1 - sigma_t) * B_h * pred
        if self.lower_order_nums < self.solver_order:
            self.lower_order_nums += 1
        self.step_index += 1
        return prev.to(sample.dtype)
