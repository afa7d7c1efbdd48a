"""File-level loaders with the reference's names and signatures (F/infer/utils_infer.py:92-130 `load_vocoder`, :175-218 `load_checkpoint`,
:224-260 `load_model`), returning the HIP-path objects (`F5HipModel`, `F5HipVocos`, `F5HipBigVGAN`) where the reference returns
`CFM` / `Vocos` / `BigVGAN` modules.  A caller of the reference (`F/infer/infer_cli.py:163`, `infer_gradio.py`) keeps its three calls:

    vocoder = load_vocoder(vocoder_name, is_local=True, local_path=...)
    model = load_model(DiT, model_cfg, mel_spec_type=..., vocab_file=...)          # architecture + tokenizer, no weights yet
    model = load_checkpoint(model, ckpt_path, device, use_ema=True)                # or load_model(..., ckpt_path=...)

Files are read with loaders that execute nothing from the file: safetensors, or `torch.load(..., weights_only=True)`.
There is no hub download on this deployment target: `is_local=False` raises (the reference fetches from huggingface there)."""
from __future__ import annotations

import json
import os

import torch

from .model import DiTArch, F5HipModel, MMDiTArch, UNetTArch
from .tokenizer import get_tokenizer
from .vocoder import F5HipBigVGAN, F5HipVocos

# the reference's module constants these functions default to (F/infer/utils_infer.py:40-53)
n_mel_channels = 100
mel_spec_type = "vocos"
ode_method = "euler"
device = "cuda" if torch.cuda.is_available() else "cpu"

_LEGACY_MEL_KEYS = ("mel_spec.mel_stft.mel_scale.fb", "mel_spec.mel_stft.spectrogram.window")   # utils_infer.py:205-208


class _Backbone:
    """What the reference passes as `model_cls` (its nn.Module classes DiT / UNetT / MMDiT): here a tag that names the architecture."""
    arch_cls = None

    def __new__(cls, **cfg):
        return cls.arch_cls(**cfg)


class DiT(_Backbone):        # F/model/backbones/dit.py:92
    arch_cls = DiTArch


class UNetT(_Backbone):      # F/model/backbones/unett.py:96
    arch_cls = UNetTArch


class MMDiT(_Backbone):      # F/model/backbones/mmdit.py:83
    arch_cls = MMDiTArch


def read_checkpoint(ckpt_path: str, use_ema: bool = True, map_location="cpu") -> dict:
    """The state_dict `load_checkpoint` hands to `model.load_state_dict` (utils_infer.py:189-215): `.safetensors` or a weights-only `.pt`;
    with `use_ema` the `ema_model_state_dict` entry, its `ema_model.` prefixes stripped, without `initted` / `step` and the two legacy
    mel-spectrogram buffers; otherwise `model_state_dict` as it stands."""
    ckpt_type = ckpt_path.split(".")[-1]
    if ckpt_type == "safetensors":
        from safetensors.torch import load_file
        checkpoint = load_file(ckpt_path, device=str(map_location))
    else:
        checkpoint = torch.load(ckpt_path, map_location=map_location, weights_only=True)
    if use_ema:
        if ckpt_type == "safetensors":
            checkpoint = {"ema_model_state_dict": checkpoint}
        sd = {k.replace("ema_model.", ""): v for k, v in checkpoint["ema_model_state_dict"].items() if k not in ("initted", "step")}
        for key in _LEGACY_MEL_KEYS:
            sd.pop(key, None)
        return sd
    if ckpt_type == "safetensors":
        checkpoint = {"model_state_dict": checkpoint}
    return checkpoint["model_state_dict"]


class UnloadedModel:
    """`load_model` without a checkpoint: the architecture, tokenizer and sampler settings of the model-to-be.  The reference returns a
    randomly initialised CFM at this point (its `load_checkpoint` call inside `load_model` is commented out, utils_infer.py:258); sampling
    from either is meaningless, so this object refuses to."""

    def __init__(self, arch, vocab_char_map, mel_spec_type, odeint_kwargs, gemm_planes, device):
        self.arch, self.vocab_char_map, self.mel_spec_type = arch, vocab_char_map, mel_spec_type
        self.odeint_kwargs, self.gemm_planes, self.device = odeint_kwargs, gemm_planes, device

    def build(self, state_dict: dict, device=None):
        return F5HipModel(self.arch, state_dict, vocab_char_map=self.vocab_char_map, gemm_planes=self.gemm_planes,
                          device=_hip_device(device if device is not None else self.device), mel_spec_type=self.mel_spec_type,
                          odeint_kwargs=self.odeint_kwargs)

    def sample(self, *a, **k):
        raise RuntimeError("this model has no weights yet: call load_checkpoint(model, ckpt_path, device) (or load_model(..., ckpt_path=...))")

    sample_units = transformer_forward = sample

    def to(self, device):
        self.device = device
        return self

    def eval(self):
        return self


def _hip_device(dev):
    dev = torch.device(dev)
    return torch.device("cuda:0") if dev.type == "cuda" and dev.index is None else dev


def load_checkpoint(model, ckpt_path, device: str, dtype=None, use_ema=True):
    """utils_infer.py:175-218.  `model`: what `load_model` returned (or an `F5HipModel`, whose architecture and settings are kept);
    returns the model carrying the file's weights.  `dtype` is accepted for signature parity: the reference forces fp32 here
    (utils_infer.py:176-184) and the HIP path has its own operand precisions (`gemm_planes`)."""
    if not isinstance(model, (UnloadedModel, F5HipModel)):
        raise TypeError(f"load_checkpoint: expected the object load_model returned, got {type(model).__name__}")
    sd = read_checkpoint(ckpt_path, use_ema=use_ema)
    if isinstance(model, UnloadedModel):
        return model.build(sd, device)
    return F5HipModel(model.arch, sd, vocab_char_map=model.vocab_char_map, gemm_planes=model.gemm_planes, device=_hip_device(device),
                      mel_spec_type=model.mel_spec_type, odeint_kwargs=model.odeint_kwargs)


def load_model(model_cls, model_cfg, mel_spec_type=mel_spec_type, vocab_file="", ode_method=ode_method, use_ema=True, device=device,
               ckpt_path="", gemm_planes=3):
    """utils_infer.py:224-260: tokenizer from `vocab_file` ("custom": one token per line), backbone `model_cls(**model_cfg,
    text_num_embeds=vocab_size, mel_dim=100)`, ODE method.  `ckpt_path` (not in the reference's current signature; its callers still pass
    one positionally after `model_cfg`) loads the weights at once; without it the result goes through `load_checkpoint`."""
    if vocab_file == "":
        raise ValueError("load_model: vocab_file is required (the reference's packaged infer/examples/vocab.txt is not shipped here)")
    vocab_char_map, vocab_size = get_tokenizer(vocab_file, "custom")
    arch = model_cls(**model_cfg, text_num_embeds=vocab_size, mel_dim=n_mel_channels)
    model = UnloadedModel(arch, vocab_char_map, mel_spec_type, dict(method=ode_method), gemm_planes, device)
    if ckpt_path:
        return load_checkpoint(model, ckpt_path, device, use_ema=use_ema)
    return model


def _vocos_hparams(config_path: str) -> dict:
    """The constructor arguments of vocos 0.1.0's config.yaml (feature_extractor / backbone / head `init_args`) that the decoder needs."""
    import yaml
    with open(config_path, "r", encoding="utf-8") as f:
        cfg = yaml.safe_load(f)
    bb = cfg["backbone"]["init_args"]
    head = cfg["head"]["init_args"]
    return dict(in_channels=bb["input_channels"], dim=bb["dim"], intermediate_dim=bb["intermediate_dim"], num_layers=bb["num_layers"],
                n_fft=head["n_fft"], hop_length=head["hop_length"])


def load_vocoder(vocoder_name="vocos", is_local=False, local_path="", device=device, hf_cache_dir=None):
    """utils_infer.py:92-130.  vocos: `{local_path}/config.yaml` + `pytorch_model.bin`; bigvgan: `{local_path}/config.json` +
    `bigvgan_generator.pt` (BigVGAN v2's `from_pretrained` layout: `{"generator": state_dict}` with weight-norm parameters, folded here
    like `remove_weight_norm()`)."""
    if not is_local:
        raise RuntimeError(f"load_vocoder({vocoder_name!r}, is_local=False): no hub download on this target; pass is_local=True and local_path")
    dev = _hip_device(device)
    if vocoder_name == "vocos":
        hp = _vocos_hparams(os.path.join(local_path, "config.yaml"))
        sd = torch.load(os.path.join(local_path, "pytorch_model.bin"), map_location="cpu", weights_only=True)
        return F5HipVocos(sd, device=dev, **hp)
    if vocoder_name == "bigvgan":
        with open(os.path.join(local_path, "config.json"), "r", encoding="utf-8") as f:
            h = json.load(f)
        ck = torch.load(os.path.join(local_path, "bigvgan_generator.pt"), map_location="cpu", weights_only=True)
        sd = ck["generator"] if "generator" in ck else ck
        return F5HipBigVGAN(sd, num_mels=h["num_mels"], upsample_rates=tuple(h["upsample_rates"]),
                            upsample_kernel_sizes=tuple(h["upsample_kernel_sizes"]), upsample_initial_channel=h["upsample_initial_channel"],
                            resblock_kernel_sizes=tuple(h["resblock_kernel_sizes"]),
                            resblock_dilation_sizes=tuple(tuple(d) for d in h["resblock_dilation_sizes"]), device=dev)
    raise ValueError(f"unknown vocoder {vocoder_name!r}")
