"""Flow-UniPC: oracle vs the reference's own scheduler file (golden), host coefficient table vs oracle
(CPU emulation of the fused update formula), and the HIP fused CFG+UniPC kernel vs oracle (gpu)."""
import os

import numpy as np
import pytest
import torch

from oracle.unipc_oracle import UniPCOracle
from oracle.gen_golden_unipc import synthetic_velocity

CASES = ["n50_s5", "n8_s2", "n4_s5"]


def _run_oracle(fx, dtype=torch.float32):
    o = UniPCOracle()
    o.set_timesteps(fx["n"], shift=fx["shift"])
    x = torch.randn(fx["shape"], generator=torch.Generator().manual_seed(42)).to(dtype)
    traj = []
    for i in range(fx["n"]):
        v = (0.3 * x + synthetic_velocity(i, fx["shape"]).to(dtype))
        x = o.step(v, x)
        traj.append(x.clone())
    return o, torch.stack(traj)


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_scheduler_golden(golden_dir, name):
    fx = torch.load(os.path.join(golden_dir, f"unipc_{name}.pt"))
    o, traj = _run_oracle(fx)
    assert torch.equal(o.timesteps, fx["timesteps"])
    assert torch.equal(o.sigmas, fx["sigmas"])
    assert torch.allclose(traj, fx["traj"], rtol=0, atol=2e-6), (traj - fx["traj"]).abs().max()


@pytest.mark.parametrize("name", CASES)
def test_coefficient_table_reproduces_reference_trajectory(golden_dir, name):
    """x_next = p0*xc + p1*x0 + p2*m0 with xc = a0*x_last + a1*m0 + a2*m1 + a3*x0 — the exact formula
    ce_cfg_unipc_step evaluates — replayed in fp64 numpy with the host table."""
    from chronoedit_amd.scheduler import FlowUniPCMultistepScheduler
    fx = torch.load(os.path.join(golden_dir, f"unipc_{name}.pt"))
    s = FlowUniPCMultistepScheduler(flow_shift=fx["shift"])
    s.set_timesteps(fx["n"])
    assert torch.equal(s.timesteps, fx["timesteps"])
    assert torch.allclose(s.sigmas, fx["sigmas"], atol=1e-7)
    x = torch.randn(fx["shape"], generator=torch.Generator().manual_seed(42)).double().numpy()
    m0 = np.zeros_like(x)
    m1 = np.zeros_like(x)
    x_last = np.zeros_like(x)
    for i in range(fx["n"]):
        c = s.coef[i]
        v = 0.3 * x + synthetic_velocity(i, fx["shape"]).double().numpy()
        x0 = x - c[1] * v
        xc = c[3] * x_last + c[4] * m0 + c[5] * m1 + c[6] * x0 if c[2] else x
        xn = c[7] * xc + c[8] * x0 + c[9] * m0
        x_last, m1, m0, x = xc, m0, x0, xn
        ref = fx["traj"][i].double().numpy()
        assert np.abs(x - ref).max() < 2e-5 * max(1.0, np.abs(ref).max()), (i, np.abs(x - ref).max())


def test_diffusers_style_grid_equals_sibling_grid():
    from chronoedit_amd.scheduler import flow_sigmas
    for n, sh in [(50, 5.0), (8, 2.0)]:
        a, b = flow_sigmas(n, sh, grid="sibling"), flow_sigmas(n, sh, grid="diffusers")
        assert np.abs(a - b).max() < 1e-6
        assert np.array_equal((a * 1000).astype(np.int64), (b * 1000).astype(np.int64)) or np.abs(a - b).max() < 1e-7


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_fused_step_matches_reference_trajectory(golden_dir, name):
    from chronoedit_amd.scheduler import FlowUniPCMultistepScheduler
    fx = torch.load(os.path.join(golden_dir, f"unipc_{name}.pt"))
    dev = torch.device("cuda:0")
    s = FlowUniPCMultistepScheduler(flow_shift=fx["shift"])
    s.set_timesteps(fx["n"], device=dev)
    x = torch.randn(fx["shape"], generator=torch.Generator().manual_seed(42)).to(dev)
    # the model output is bf16 on the product path: feed bf16-representable velocities to both sides
    o = UniPCOracle()
    o.set_timesteps(fx["n"], shift=fx["shift"])
    xo = x.cpu().clone()
    for i, t in enumerate(s.timesteps):
        v = (0.3 * xo + synthetic_velocity(i, fx["shape"])).to(torch.bfloat16)
        xo = o.step(v.float(), xo)
        x = s.step(v.to(dev), t, x, return_dict=False)[0]
        err = (x.cpu() - xo).abs().max().item()
        assert err < 1e-4 * max(1.0, xo.abs().max().item()), (i, err)


@pytest.mark.gpu
def test_hip_cfg_combine_rounding():
    """noise = u + g*(c-u) in bf16 tensor arithmetic (pipeline_chronoedit.py:736), then one UniPC step."""
    from chronoedit_amd.scheduler import FlowUniPCMultistepScheduler
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    shape = (1, 16, 2, 8, 8)
    c = torch.randn(shape, generator=g).to(torch.bfloat16)
    u = torch.randn(shape, generator=g).to(torch.bfloat16)
    x = torch.randn(shape, generator=g)
    s = FlowUniPCMultistepScheduler(flow_shift=5.0)
    s.set_timesteps(4, device=dev)
    out = s.step_cfg(c.to(dev), u.to(dev), 5.0, x.to(dev).clone())
    noise = u + 5.0 * (c - u)  # bf16 ops, rounding at each
    o = UniPCOracle()
    o.set_timesteps(4, shift=5.0)
    ref = o.step(noise.float(), x)
    assert (out.cpu() - ref).abs().max().item() < 1e-5 * max(1.0, ref.abs().max().item())
