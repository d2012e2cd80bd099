"""TEST INFRASTRUCTURE - stand-ins for the diffusers / imaginaire names that the reference's own pipeline source
(/root/reference/chronoedit_diffusers/pipeline_chronoedit.py) refers to, so that `oracle/build_ref.py` can lift the reference's
`__call__`, `prepare_latents`, `encode_prompt`, `encode_image`, `check_inputs` and property sources VERBATIM into
`oracle/_ref/pipeline_ref.bin` (a compiled code object generated at build time, never committed) and run them over the chronoedit_amd drop-ins.

Nothing here restates reference logic: these are the un-vendored leaves (diffusers==0.35.2 is not installable here) - a logger,
a progress bar, `randn_tensor`, `VideoProcessor` pre / post processing, the output dataclass.  Only tests/ may import this."""
from __future__ import annotations

import contextlib
from dataclasses import dataclass
from typing import Any, Callable, Dict, List, Optional, Tuple, Union  # noqa: F401  (names used by the lifted source)

import numpy as np  # noqa: F401
import PIL  # noqa: F401
import PIL.Image  # noqa: F401
import torch

XLA_AVAILABLE = False
EXAMPLE_DOC_STRING = ""
PipelineImageInput = Any


class PipelineCallback:  # diffusers.callbacks
    tensor_inputs: List[str] = []


class MultiPipelineCallbacks(PipelineCallback):
    pass


def replace_example_docstring(_doc):
    return lambda fn: fn


class _Log:
    def __getattr__(self, name):
        return lambda *a, **k: None


logger = _Log()
log = _Log()


class _GuardrailPresets:
    @staticmethod
    def run_text_guardrail(prompt, runner):
        return runner(prompt)

    @staticmethod
    def run_video_guardrail(frames, runner):
        return runner(frames)


guardrail_presets = _GuardrailPresets()


@dataclass
class WanPipelineOutput:
    frames: Any


def randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
    from chronoedit_amd.pipeline import _randn_tensor
    return _randn_tensor(tuple(shape), generator, device, dtype)


class _VideoProcessor:
    """diffusers.video_processor.VideoProcessor: the two calls the reference makes (pipeline_chronoedit.py:673,801)."""

    def preprocess(self, image, height=None, width=None):
        from chronoedit_amd.pipeline import ChronoEditPipeline
        return ChronoEditPipeline.preprocess_image(image, height, width)

    def postprocess_video(self, video, output_type="np"):
        from chronoedit_amd.pipeline import ChronoEditPipeline
        return ChronoEditPipeline.postprocess_video(video, output_type)


class RefHarnessBase:
    """What DiffusionPipeline gives the reference class: component registration, `_execution_device`, a progress bar, hooks."""

    _callback_tensor_inputs = ["latents", "prompt_embeds", "negative_prompt_embeds"]

    def __init__(self, tokenizer=None, text_encoder=None, image_encoder=None, image_processor=None, transformer=None, vae=None,
                 scheduler=None):
        self.tokenizer, self.text_encoder, self.image_encoder, self.image_processor = tokenizer, text_encoder, image_encoder, image_processor
        self.transformer, self.vae, self.scheduler = transformer, vae, scheduler
        self.vae_scale_factor_temporal = 2 ** sum(self.vae.temperal_downsample) if getattr(self, "vae", None) else 4
        self.vae_scale_factor_spatial = 2 ** len(self.vae.temperal_downsample) if getattr(self, "vae", None) else 8
        self.video_processor = _VideoProcessor()
        self.text_guardrail_runner = None
        self.video_guardrail_runner = None

    @property
    def _execution_device(self):
        return torch.device(self.transformer.device)

    @contextlib.contextmanager
    def progress_bar(self, total=None):
        class _Bar:
            def update(self, *a):
                pass
        yield _Bar()

    def maybe_free_model_hooks(self):
        pass
