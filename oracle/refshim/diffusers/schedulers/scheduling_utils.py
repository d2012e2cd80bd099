from dataclasses import dataclass
from enum import Enum

import torch


class KarrasDiffusionSchedulers(Enum):
    UniPCMultistepScheduler = 1


class SchedulerMixin:
    pass


@dataclass
class SchedulerOutput:
    prev_sample: "torch.Tensor"
