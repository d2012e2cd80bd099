from dataclasses import dataclass

import torch


@dataclass
class Transformer2DModelOutput:
    sample: "torch.Tensor"
