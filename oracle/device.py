"""Where the -m gpu tests EVALUATE the fp32 oracle (test infrastructure, like everything under oracle/: nothing in chronoedit_amd/ imports it).

The oracle (dit_oracle / pipeline_oracle / vae_oracle) is plain torch; at the 14B width its fp32 evaluation on the host cores is what the GPU
suite spends most of its wall time on (13 GB of fp32 weights, 50 s for the configs[0] edit at L = 4).  The full-width tests therefore run the
SAME oracle functions with their tensors on the device - torch's fp32 kernels, no reduced-precision mode (torch.backends.cuda.matmul.allow_tf32
is off, checked below) - and compare on the host.  CE_ORACLE_DEVICE=cpu puts every evaluation back on the host cores (the two agree to ~1e-6
rel-L2, three orders below any bound in the suite); bench.py's cpu_baseline leg and the -m "not gpu" tests always run on the host."""
import contextlib
import os

import torch


def oracle_device():
    want = os.environ.get("CE_ORACLE_DEVICE", "cuda")
    if want.startswith("cuda") and torch.cuda.is_available():
        assert not torch.backends.cuda.matmul.allow_tf32, "the fp32 oracle must not run in a reduced-precision matmul mode"
        return torch.device("cuda:0")
    return torch.device("cpu")


@contextlib.contextmanager
def on(dev=None):
    """with on() as dev: the oracle's own factory calls (torch.arange / zeros / tensor without a device) land on `dev`."""
    dev = oracle_device() if dev is None else torch.device(dev)
    if dev.type == "cpu":
        yield dev
    else:
        with torch.device(dev):
            yield dev


def to(tree, dev, dtype=None):
    """tensors of a dict / list / tuple (or one tensor) on `dev` (and in `dtype`, converted on the device); anything else unchanged"""
    if torch.is_tensor(tree):
        t = tree.to(dev)
        return t.to(dtype) if dtype is not None and t.is_floating_point() else t
    if isinstance(tree, dict):
        return {k: to(v, dev, dtype) for k, v in tree.items()}
    if isinstance(tree, (list, tuple)):
        return type(tree)(to(v, dev, dtype) for v in tree)
    return tree
