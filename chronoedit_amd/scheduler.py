"""Flow-matching UniPC multistep scheduler for the `pipe.scheduler` slot, MI355X-first.

Interface parity with what ChronoEditPipeline touches (SURVEY.md §8b "Scheduler slot";
pipeline_chronoedit.py:668-669,704-709,739; run_inference_diffusers.py:379-382):
`from_config(cfg, flow_shift=)`, `.config`, `.order`, `.set_timesteps(n, device=)`, `.timesteps`,
`.sigmas`, `.step(model_output, t, sample, return_dict=False)[0]`, and the MUTABLE history
`.model_outputs` (list) / `.last_sample` that the pipeline slices at the temporal-reasoning
truncation.

MI355X-first design: all scalar work of the reference (`lambda/h/phi/rho`, fm_solvers_unipc.py:
420-468,560-620 — host `.item()`-style math every step) is done ONCE in `set_timesteps`, in float64,
into a [n_steps, 10] coefficient table resident on the device.  A step is then one fused HIP
launch (`ce_cfg_unipc_step`) that also applies classifier-free guidance — no host<->device sync,
so the whole denoising step is hipGraph-capturable.  Latents and history are kept in fp32.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import List, Optional, Tuple

import numpy as np
import torch

from . import ops


def _lam(sigma: float) -> float:
    if sigma <= 0.0:
        return math.inf
    return math.log1p(-sigma) - math.log(sigma)


def _bh2_terms(sigma_t: float, sigma_s0: float, order: int, rk_sigmas: List[float]):
    """rks, b (length `order`), h_phi_1, B_h exactly as fm_solvers_unipc.py:417-468 (predict_x0, bh2)."""
    lam_t, lam_s0 = _lam(sigma_t), _lam(sigma_s0)
    h = lam_t - lam_s0
    rks = [(_lam(s) - lam_s0) / h for s in rk_sigmas] + [1.0]
    hh = -h
    h_phi_1 = math.expm1(hh)
    h_phi_k = (h_phi_1 / hh - 1.0) if math.isfinite(hh) else -1.0
    B_h = h_phi_1
    b, fact = [], 1
    for i in range(1, order + 1):
        b.append(h_phi_k * fact / B_h)
        fact *= i + 1
        h_phi_k = (h_phi_k / hh if math.isfinite(hh) else 0.0) - 1.0 / fact
    return rks, b, h_phi_1, B_h


def unipc_coefficients(sigmas: np.ndarray, solver_order: int = 2, lower_order_final: bool = True,
                       disable_corrector: Tuple[int, ...] = ()) -> np.ndarray:
    """[n, 10] float64 rows {g(unused=0), sigma_i, use_corr, a0..a3, p0..p2} for ce_cfg_unipc_step.

    Corrector (fm_solvers_unipc.py:501-641) at step i with the order of step i-1's predictor:
        xc = a0*x_last + a1*m0 + a2*m1 + a3*x0
    Predictor (:365-499) with this step's order (history already shifted, so m0_new = x0):
        x_next = p0*xc + p1*x0 + p2*m0_old
    """
    assert solver_order in (1, 2), "the fused update covers solver_order <= 2 (reference default 2)"
    n = len(sigmas) - 1
    rows = np.zeros((n, 10), dtype=np.float64)
    prev_order = None
    for i in range(n):
        s_i = float(sigmas[i])
        rows[i, 1] = s_i
        lower_order_nums = min(i, solver_order)
        # ---- corrector
        if i > 0 and (i - 1) not in disable_corrector:
            order = prev_order
            sigma_t, sigma_s0 = s_i, float(sigmas[i - 1])
            alpha_t = 1.0 - sigma_t
            rk_s = [float(sigmas[i - 2])] if order == 2 else []
            rks, b, h_phi_1, B_h = _bh2_terms(sigma_t, sigma_s0, order, rk_s)
            if order == 1:
                rhos = [0.5]
            else:
                R = np.array([[1.0, 1.0], [rks[0], 1.0]])
                rhos = list(np.linalg.solve(R, np.array(b)))
            a0 = sigma_t / sigma_s0
            a1 = -alpha_t * h_phi_1 + alpha_t * B_h * rhos[-1]
            a2 = 0.0
            if order == 2:
                a1 += alpha_t * B_h * rhos[0] / rks[0]
                a2 = -alpha_t * B_h * rhos[0] / rks[0]
            a3 = -alpha_t * B_h * rhos[-1]
            rows[i, 2:7] = [1.0, a0, a1, a2, a3]
        # ---- predictor
        order = min(solver_order, n - i) if lower_order_final else solver_order
        order = min(order, lower_order_nums + 1)
        prev_order = order
        sigma_t, sigma_s0 = float(sigmas[i + 1]), s_i
        alpha_t = 1.0 - sigma_t
        rk_s = [float(sigmas[i - 1])] if order == 2 else []
        rks, b, h_phi_1, B_h = _bh2_terms(sigma_t, sigma_s0, order, rk_s)
        p0 = sigma_t / sigma_s0
        p1 = -alpha_t * h_phi_1
        p2 = 0.0
        if order == 2:
            p1 += alpha_t * B_h * 0.5 / rks[0]
            p2 = -alpha_t * B_h * 0.5 / rks[0]
        rows[i, 7:10] = [p0, p1, p2]
    return rows


def flow_sigmas(n: int, shift: float, num_train_timesteps: int = 1000, grid: str = "sibling") -> np.ndarray:
    """Shifted flow sigma grid of length n (fm_solvers_unipc.py:196-208).  grid="diffusers" builds it the way
    diffusers 0.35.2 `use_flow_sigmas` does (linspace on alphas; same values up to fp rounding)."""
    T = num_train_timesteps
    if grid == "sibling":
        alphas = np.linspace(1, 1 / T, T)[::-1].copy()
        base = (1.0 - alphas).astype(np.float32)
        sig = np.linspace(float(base[0]), float(base[-1]), n + 1).copy()[:-1]
    else:
        sig = (1.0 - np.linspace(1, 1 / T, n + 1))[::-1].copy()[:-1]
    return shift * sig / (1 + (shift - 1) * sig)


class FlowUniPCMultistepScheduler:
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, solver_order: int = 2, prediction_type: str = "flow_prediction",
                 shift: float = 1.0, flow_shift: Optional[float] = None, predict_x0: bool = True, solver_type: str = "bh2",
                 lower_order_final: bool = True, disable_corrector: Tuple[int, ...] = (), final_sigmas_type: str = "zero",
                 use_flow_sigmas: bool = True, sigma_grid: str = "sibling", **unused):
        if prediction_type != "flow_prediction" or not predict_x0 or solver_type != "bh2" or final_sigmas_type != "zero":
            raise NotImplementedError("only the ChronoEdit configuration (flow_prediction, predict_x0, bh2, final sigma 0) is built")
        if flow_shift is not None:  # diffusers spelling used by run_inference_diffusers.py:379-382
            shift = flow_shift
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, solver_order=solver_order,
                                      prediction_type=prediction_type, shift=shift, flow_shift=shift, predict_x0=predict_x0,
                                      solver_type=solver_type, lower_order_final=lower_order_final,
                                      disable_corrector=tuple(disable_corrector), final_sigmas_type=final_sigmas_type,
                                      use_flow_sigmas=use_flow_sigmas, sigma_grid=sigma_grid)
        self.num_inference_steps = None
        self.timesteps = None
        self.sigmas = None
        self.model_outputs: List[Optional[torch.Tensor]] = [None] * solver_order
        self.last_sample: Optional[torch.Tensor] = None
        self._step_index = None
        self._coef_dev = None
        # torch.float32 (default): latents and history keep full fp32 values for the whole trajectory.  torch.bfloat16: they are
        # rounded to bf16 values after every step, the rounding points of the reference, whose latents / model_outputs /
        # last_sample are bf16 tensors (pipeline_chronoedit.py:681,739) - "reference-precision trajectory" mode.
        self.trajectory_dtype = torch.float32

    @classmethod
    def from_config(cls, config, **overrides):
        d = dict(vars(config)) if not isinstance(config, dict) else dict(config)
        d.update(overrides)
        if "flow_shift" in overrides:
            d["shift"] = overrides["flow_shift"]
            d["flow_shift"] = None
        return cls(**d)

    @property
    def step_index(self):
        return self._step_index

    def set_timesteps(self, num_inference_steps: int, device=None, shift: Optional[float] = None):
        c = self.config
        if shift is None:
            shift = c.shift
        sig = flow_sigmas(num_inference_steps, shift, c.num_train_timesteps, c.sigma_grid)
        self.timesteps = torch.from_numpy(sig * c.num_train_timesteps).to(device=device, dtype=torch.int64)
        sig_full = np.concatenate([sig, [0.0]]).astype(np.float32)
        self.sigmas = torch.from_numpy(sig_full)  # stays on the host like the reference (:240)
        self.num_inference_steps = num_inference_steps
        self.coef = unipc_coefficients(sig_full.astype(np.float64), c.solver_order, c.lower_order_final, c.disable_corrector)
        self._coef_dev = None if device is None else torch.from_numpy(self.coef.astype(np.float32)).to(device)
        self._device = device
        self.model_outputs = [None] * c.solver_order
        self.last_sample = None
        self._step_index = None
        self._g_cache = {}

    def coef_row(self, i: int, guidance_scale: float, device) -> torch.Tensor:
        """Device view of step i's coefficient row (guidance scale in column 0)."""
        return self._coef_row(i, guidance_scale, device)

    # -- state helpers -----------------------------------------------------------------
    def _ensure_state(self, like: torch.Tensor):
        """History buffers (fp32, latent-shaped).  Re-created when the pipeline sliced them to a new frame count."""
        for j in range(len(self.model_outputs)):
            t = self.model_outputs[j]
            if t is None or t.shape != like.shape:
                self.model_outputs[j] = torch.zeros(like.shape, dtype=torch.float32, device=like.device) if t is None else t.to(torch.float32).contiguous()
            elif t.dtype != torch.float32 or not t.is_contiguous():
                self.model_outputs[j] = t.to(torch.float32).contiguous()
        if self.last_sample is None:
            self.last_sample = torch.zeros(like.shape, dtype=torch.float32, device=like.device)
        elif self.last_sample.dtype != torch.float32 or not self.last_sample.is_contiguous():
            self.last_sample = self.last_sample.to(torch.float32).contiguous()

    def _coef_row(self, i: int, guidance_scale: float, device) -> torch.Tensor:
        if self._coef_dev is None or self._coef_dev.device != device:
            self._coef_dev = torch.from_numpy(self.coef.astype(np.float32)).to(device)
        g = float(guidance_scale)
        if self._g_cache.get("g") != g:  # guidance scale lives in column 0 of every row
            self._coef_dev[:, 0] = g
            self._g_cache["g"] = g
        return self._coef_dev[i]

    # -- fused fast path -----------------------------------------------------------------
    def step_cfg(self, v_cond: torch.Tensor, v_uncond: Optional[torch.Tensor], guidance_scale: float, sample: torch.Tensor,
                 coef: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One loop tail (pipeline_chronoedit.py:736-739) as a single HIP launch.  `sample` must be fp32 and
        is updated IN PLACE and returned.  `coef`: explicit device coefficient row (hipGraph replay feeds the row of the
        current step through one fixed buffer); default = this step's row of the table."""
        if self._step_index is None:
            self._step_index = 0
        i = self._step_index
        if coef is None and i >= self.num_inference_steps:
            raise IndexError("scheduler stepped past the last timestep")
        assert sample.dtype == torch.float32 and sample.is_contiguous()
        self._ensure_state(sample)
        m1, m0 = (self.model_outputs[0], self.model_outputs[1]) if len(self.model_outputs) == 2 else (self.model_outputs[0], self.model_outputs[0])
        if len(self.model_outputs) == 1:
            m1 = torch.zeros_like(m0)
        if coef is None:
            coef = self._coef_row(i, guidance_scale, sample.device)
        ops.cfg_unipc_step(v_cond.to(torch.bfloat16).contiguous(), None if v_uncond is None else v_uncond.to(torch.bfloat16).contiguous(),
                           sample, self.last_sample, m0, m1, coef, round_sigma_v=self.trajectory_dtype == torch.bfloat16,
                           bf16_state=self.trajectory_dtype == torch.bfloat16)
        self._step_index = i + 1
        return sample

    # -- reference-compatible API --------------------------------------------------------
    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, return_dict: bool = True, generator=None):
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        x = sample.to(torch.float32).contiguous()
        if x.data_ptr() == sample.data_ptr():
            x = x.clone()  # the reference returns a new tensor
        self.step_cfg(model_output, None, 1.0, x)
        prev = x.to(sample.dtype)
        if not return_dict:
            return (prev, self.model_outputs[-1])
        return SimpleNamespace(prev_sample=prev)

    def scale_model_input(self, sample, *a, **k):
        return sample
