"""TEST INFRASTRUCTURE - stand-ins for the diffusers / imaginaire names that the reference's own pipeline source
(/root/reference/chronoedit_diffusers/pipeline_chronoedit.py) refers to, so that `oracle/build_ref.py` can lift the reference's
`__call__`, `prepare_latents`, `encode_prompt`, `encode_image`, `check_inputs` and property sources VERBATIM into
`oracle/_ref/pipeline_ref.bin` (a compiled code object generated at build time, never committed) and run them over the chronoedit_amd drop-ins.

Nothing here restates reference logic: these are the un-vendored leaves (diffusers==0.35.2 is not installable here) - a logger,
a progress bar, `randn_tensor`, `VideoProcessor` pre / post processing, the output dataclass - each written here on its own (plain
torch / PIL restatements of the diffusers behaviour), NOT delegated to chronoedit_amd, so that tests/test_ref_loop_gpu.py compares
the engine pipeline's noise / pre- / post-processing against an independent statement.  Only tests/ may import this."""
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
    """diffusers.utils.torch_utils.randn_tensor restated from its 0.35.2 behaviour (NOT delegated to chronoedit_amd, so that the
    reference-loop test can see a divergence in how the engine draws its noise): the noise is drawn on the generator's device -
    a CPU generator with a GPU target draws on the CPU and moves the result; a list of generators draws one sample each."""
    shape = tuple(shape)
    device = torch.device(device) if device is not None else torch.device("cpu")
    draw_on = device
    gens = generator if isinstance(generator, (list, tuple)) else None
    first = gens[0] if gens else generator
    if first is not None:
        gtype = first.device.type
        if gtype != device.type:
            if gtype == "cpu":
                draw_on = torch.device("cpu")
            else:
                raise ValueError(f"Cannot generate a {device.type} tensor from a generator of type {gtype}.")
    if gens is not None and len(gens) == 1:
        generator, gens = gens[0], None
    if gens is not None:
        one = (1,) + shape[1:]
        return torch.cat([torch.randn(one, generator=g, device=draw_on, dtype=dtype) for g in gens], dim=0).to(device)
    return torch.randn(shape, generator=generator, device=draw_on, dtype=dtype).to(device)


class _VideoProcessor:
    """diffusers.video_processor.VideoProcessor restated from its 0.35.2 behaviour for the two calls the reference makes
    (pipeline_chronoedit.py:673,801) - independently of chronoedit_amd.pipeline's own pre / post processing, which the
    reference-loop test compares against: PIL images are resized with Lanczos and scaled to [0, 1]; arrays / tensors are resized with
    F.interpolate's default (nearest); everything is mapped to [-1, 1] unless it already carries negative values."""

    def preprocess(self, image, height=None, width=None):
        if isinstance(image, PIL.Image.Image):
            image = [image]
        if isinstance(image, (list, tuple)) and isinstance(image[0], PIL.Image.Image):
            arrs = [np.asarray(im.convert("RGB").resize((width, height), resample=PIL.Image.LANCZOS)).astype(np.float32) / 255.0 for im in image]
            x = torch.from_numpy(np.stack(arrs, axis=0)).permute(0, 3, 1, 2)
        elif isinstance(image, np.ndarray):
            x = torch.from_numpy(image if image.ndim == 4 else image[None]).float().permute(0, 3, 1, 2)
            if x.shape[-2:] != (height, width):
                x = torch.nn.functional.interpolate(x, size=(height, width))
        else:
            x = image.float()
            x = x if x.dim() == 4 else x[None]
            if x.shape[-2:] != (height, width):
                x = torch.nn.functional.interpolate(x, size=(height, width))
        if x.min() < 0:  # already in [-1, 1]: diffusers warns and skips the normalisation
            return x
        return 2.0 * x - 1.0

    def postprocess_video(self, video, output_type="np"):
        outs = []
        for sample in video:  # [3, F, H, W] -> frames [F, 3, H, W] -> [0, 1]
            frames = (sample.permute(1, 0, 2, 3).float() / 2 + 0.5).clamp(0, 1)
            if output_type == "pt":
                outs.append(frames)
                continue
            arr = frames.cpu().permute(0, 2, 3, 1).numpy()
            outs.append(arr if output_type == "np" else [PIL.Image.fromarray((f * 255).round().astype("uint8")) for f in arr])
        if output_type == "np":
            return np.stack(outs)
        if output_type == "pt":
            return torch.stack(outs)
        if output_type == "pil":
            return outs
        raise ValueError(f"{output_type} does not exist. Please choose one of ['np', 'pt', 'pil']")


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


# ---------------------------------------------------------------------------------------------------------------------
# scripts/run_inference_diffusers.py: the import names of the reference runner bound to the drop-ins
# ---------------------------------------------------------------------------------------------------------------------
@contextlib.contextmanager
def runner_shims(exported_videos: Optional[list] = None):
    """While active, the module names scripts/run_inference_diffusers.py imports (:64-80) resolve to the chronoedit_amd drop-ins -
    the edit INTEGRATION.md section 1 asks a maintainer to make, done here in `sys.modules` so that the reference's script runs
    UNMODIFIED (its compiled code, oracle/build_ref.build_runner):
        diffusers.AutoencoderKLWan                      -> chronoedit_amd.vae.AutoencoderKLWan
        diffusers.schedulers.UniPCMultistepScheduler    -> chronoedit_amd.scheduler.FlowUniPCMultistepScheduler
        diffusers.utils.export_to_video / load_image    -> frame recorder (appends to `exported_videos`) / PIL loader
        transformers.CLIPVisionModel                    -> chronoedit_amd.clip_vision.CLIPVisionModel (everything else: the real package)
        chronoedit_diffusers.pipeline_chronoedit / .transformer_chronoedit -> the chronoedit_amd classes of the same names
        scripts.prompt_enhancer                         -> refuses (out of scope: SURVEY section 2)"""
    import sys
    import types

    import transformers as real_transformers

    from chronoedit_amd.clip_vision import CLIPVisionModel
    from chronoedit_amd.pipeline import ChronoEditPipeline
    from chronoedit_amd.scheduler import FlowUniPCMultistepScheduler
    from chronoedit_amd.transformer import ChronoEditTransformer3DModel
    from chronoedit_amd.vae import AutoencoderKLWan

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        return m

    def load_image(path):
        return PIL.Image.open(path).convert("RGB")

    def export_to_video(frames, path, fps=8):
        if exported_videos is not None:
            exported_videos.append((path, fps, np.asarray(frames)))
        return path

    def refuse(*a, **k):
        raise RuntimeError("the prompt enhancer (Qwen-VL) is outside the hot path (SURVEY.md section 2)")

    class _Transformers(types.ModuleType):
        def __getattr__(self, name):  # AutoTokenizer, CLIPImageProcessor, ...: the host-side objects stay the reference's
            return getattr(real_transformers, name)

    tf = _Transformers("transformers")
    tf.CLIPVisionModel = CLIPVisionModel
    shims = {
        "diffusers": mod("diffusers", AutoencoderKLWan=AutoencoderKLWan),
        "diffusers.schedulers": mod("diffusers.schedulers", UniPCMultistepScheduler=FlowUniPCMultistepScheduler),
        "diffusers.utils": mod("diffusers.utils", export_to_video=export_to_video, load_image=load_image),
        "transformers": tf,
        "chronoedit_diffusers": mod("chronoedit_diffusers"),
        "chronoedit_diffusers.pipeline_chronoedit": mod("chronoedit_diffusers.pipeline_chronoedit", ChronoEditPipeline=ChronoEditPipeline),
        "chronoedit_diffusers.transformer_chronoedit": mod("chronoedit_diffusers.transformer_chronoedit", ChronoEditTransformer3DModel=ChronoEditTransformer3DModel),
        "scripts": mod("scripts"),
        "scripts.prompt_enhancer": mod("scripts.prompt_enhancer", load_model=refuse, enhance_prompt=refuse),
    }
    saved = {k: sys.modules.get(k) for k in shims}
    sys.modules.update(shims)
    try:
        yield shims
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
