"""Golden trajectories from the REFERENCE scheduler file itself
(/root/reference/chronoedit/_src/models/fm_solvers_unipc.py, imported with oracle/refshim standing
in for the diffusers base classes).  Writes tests/golden/unipc_*.pt.  Build-container only."""
import importlib.util
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "refshim"))

CASES = {"n50_s5": (50, 5.0), "n8_s2": (8, 2.0), "n4_s5": (4, 5.0)}


def synthetic_velocity(step: int, shape, seed=99):
    g = torch.Generator().manual_seed(seed + step)
    return torch.randn(shape, generator=g)


def main():
    spec = importlib.util.spec_from_file_location("ref_unipc", "/root/reference/chronoedit/_src/models/fm_solvers_unipc.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    shape = (1, 16, 2, 6, 10)
    for name, (n, shift) in CASES.items():
        s = mod.FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)
        s.set_timesteps(n, shift=shift)
        x = torch.randn(shape, generator=torch.Generator().manual_seed(42))
        traj = []
        for i, t in enumerate(s.timesteps):
            # a smooth-ish "model": velocity depends on the sample so errors propagate like a real run
            v = 0.3 * x + synthetic_velocity(i, shape)
            x = s.step(v, t, x, return_dict=False)[0]
            traj.append(x.clone())
        fx = {"n": n, "shift": shift, "shape": shape, "timesteps": s.timesteps.clone(), "sigmas": s.sigmas.clone(),
              "traj": torch.stack(traj), "source": "reference fm_solvers_unipc.py executed with oracle/refshim"}
        path = os.path.join(ROOT, "tests", "golden", f"unipc_{name}.pt")
        torch.save(fx, path)
        print(name, float(x.abs().mean()), os.path.getsize(path))


if __name__ == "__main__":
    main()
