"""Seeded synthetic weights and inputs.

No trained checkpoint, vocabulary for Indic scripts or reference audio is
obtainable offline, so parity tests, smoke() and bench.py all run on weights
and inputs regenerated from fixed seeds.  The state_dict uses the reference's
checkpoint key names (what F/infer/utils_infer.py:195-209 yields after
stripping ``ema_model.``), in the reference modules' ``named_parameters()``
order; tests/golden/gen_golden.py asserts that order against the reference's
own module graph.

Seeds: 1234 reference audio, 2345 text ids, 3456+i noise, 4567 DiT weights,
5678 Vocos weights.  Everything is generated on the CPU generator in fp32 so
the GPU box regenerates bit-identical tensors.
"""
from __future__ import annotations

import math

import torch

SEED_AUDIO, SEED_TEXT, SEED_NOISE, SEED_DIT, SEED_VOCOS, SEED_BIGVGAN = 1234, 2345, 3456, 4567, 5678, 6789


def dit_param_specs(dim=1024, depth=22, heads=16, ff_mult=2, text_dim=512, conv_layers=4, mel_dim=100,
                    text_num_embeds=2545, dim_head=64):
    """Ordered (name, shape, kind) list mirroring DiT.__init__ (F/model/backbones/dit.py:111-128)."""
    inner = heads * dim_head
    s = []
    p = "transformer.time_embed.time_mlp."
    s += [(p + "0.weight", (dim, 256), "linear"), (p + "0.bias", (dim,), "bias"),
          (p + "2.weight", (dim, dim), "linear"), (p + "2.bias", (dim,), "bias")]
    p = "transformer.text_embed."
    s += [(p + "text_embed.weight", (text_num_embeds + 1, text_dim), "embed")]
    for i in range(conv_layers):
        q = f"{p}text_blocks.{i}."
        s += [(q + "dwconv.weight", (text_dim, 1, 7), "conv"), (q + "dwconv.bias", (text_dim,), "bias"),
              (q + "norm.weight", (text_dim,), "gain"), (q + "norm.bias", (text_dim,), "bias"),
              (q + "pwconv1.weight", (text_dim * 2, text_dim), "linear"), (q + "pwconv1.bias", (text_dim * 2,), "bias"),
              (q + "grn.gamma", (1, 1, text_dim * 2), "grn"), (q + "grn.beta", (1, 1, text_dim * 2), "grn"),
              (q + "pwconv2.weight", (text_dim, text_dim * 2), "linear"), (q + "pwconv2.bias", (text_dim,), "bias")]
    p = "transformer.input_embed."
    s += [(p + "proj.weight", (dim, mel_dim * 2 + text_dim), "linear"), (p + "proj.bias", (dim,), "bias")]
    for j in (0, 2):
        s += [(f"{p}conv_pos_embed.conv1d.{j}.weight", (dim, dim // 16, 31), "conv"),
              (f"{p}conv_pos_embed.conv1d.{j}.bias", (dim,), "bias")]
    for i in range(depth):
        q = f"transformer.transformer_blocks.{i}."
        s += [(q + "attn_norm.linear.weight", (dim * 6, dim), "adaln"), (q + "attn_norm.linear.bias", (dim * 6,), "adaln_bias")]
        for nm in ("to_q", "to_k", "to_v"):
            s += [(f"{q}attn.{nm}.weight", (inner, dim), "linear"), (f"{q}attn.{nm}.bias", (inner,), "bias")]
        s += [(q + "attn.to_out.0.weight", (dim, inner), "linear"), (q + "attn.to_out.0.bias", (dim,), "bias")]
        s += [(q + "ff.ff.0.0.weight", (dim * ff_mult, dim), "linear"), (q + "ff.ff.0.0.bias", (dim * ff_mult,), "bias"),
              (q + "ff.ff.2.weight", (dim, dim * ff_mult), "linear"), (q + "ff.ff.2.bias", (dim,), "bias")]
    s += [("transformer.norm_out.linear.weight", (dim * 2, dim), "adaln"),
          ("transformer.norm_out.linear.bias", (dim * 2,), "adaln_bias"),
          ("transformer.proj_out.weight", (mel_dim, dim), "linear"), ("transformer.proj_out.bias", (mel_dim,), "bias")]
    return s


def unett_param_specs(dim=1024, depth=24, heads=16, ff_mult=4, mel_dim=100, text_num_embeds=2545, dim_head=64):
    """Ordered (name, shape, kind) list mirroring UNetT.__init__ (F/model/backbones/unett.py:113-162)."""
    inner = heads * dim_head
    text_dim = mel_dim
    s = []
    p = "transformer.time_embed.time_mlp."
    s += [(p + "0.weight", (dim, 256), "linear"), (p + "0.bias", (dim,), "bias"),
          (p + "2.weight", (dim, dim), "linear"), (p + "2.bias", (dim,), "bias")]
    s += [("transformer.text_embed.text_embed.weight", (text_num_embeds + 1, text_dim), "embed")]
    p = "transformer.input_embed."
    s += [(p + "proj.weight", (dim, mel_dim * 2 + text_dim), "linear"), (p + "proj.bias", (dim,), "bias")]
    for j in (0, 2):
        s += [(f"{p}conv_pos_embed.conv1d.{j}.weight", (dim, dim // 16, 31), "conv"),
              (f"{p}conv_pos_embed.conv1d.{j}.bias", (dim,), "bias")]
    for i in range(depth):
        q = f"transformer.layers.{i}."
        if i >= depth // 2:
            s += [(q + "0.weight", (dim, dim * 2), "linear")]
        s += [(q + "1.g", (dim,), "gain")]
        for nm in ("to_q", "to_k", "to_v"):
            s += [(f"{q}2.{nm}.weight", (inner, dim), "linear"), (f"{q}2.{nm}.bias", (inner,), "bias")]
        s += [(q + "2.to_out.0.weight", (dim, inner), "linear"), (q + "2.to_out.0.bias", (dim,), "bias")]
        s += [(q + "3.g", (dim,), "gain")]
        s += [(q + "4.ff.0.0.weight", (dim * ff_mult, dim), "linear"), (q + "4.ff.0.0.bias", (dim * ff_mult,), "bias"),
              (q + "4.ff.2.weight", (dim, dim * ff_mult), "linear"), (q + "4.ff.2.bias", (dim,), "bias")]
    s += [("transformer.norm_out.g", (dim,), "gain"),
          ("transformer.proj_out.weight", (mel_dim, dim), "linear"), ("transformer.proj_out.bias", (mel_dim,), "bias")]
    return s


def mmdit_param_specs(dim=512, depth=16, heads=16, ff_mult=2, mel_dim=100, text_num_embeds=256, dim_head=64):
    """Ordered (name, shape, kind) list mirroring MMDiT.named_parameters() (F/model/backbones/mmdit.py:96-113, MMDiTBlock / Attention
    F/model/modules.py:588-612, 335-390): the last block is context-pre-only (2 dim modulation for the text stream, no to_out_c, no ff_c)."""
    inner = heads * dim_head
    s = []
    p = "transformer.time_embed.time_mlp."
    s += [(p + "0.weight", (dim, 256), "linear"), (p + "0.bias", (dim,), "bias"),
          (p + "2.weight", (dim, dim), "linear"), (p + "2.bias", (dim,), "bias")]
    s += [("transformer.text_embed.text_embed.weight", (text_num_embeds + 1, dim), "embed")]
    p = "transformer.audio_embed."
    s += [(p + "linear.weight", (dim, mel_dim * 2), "linear"), (p + "linear.bias", (dim,), "bias")]
    for j in (0, 2):
        s += [(f"{p}conv_pos_embed.conv1d.{j}.weight", (dim, dim // 16, 31), "conv"),
              (f"{p}conv_pos_embed.conv1d.{j}.bias", (dim,), "bias")]
    for i in range(depth):
        last = i == depth - 1
        q = f"transformer.transformer_blocks.{i}."
        s += [(q + "attn_norm_c.linear.weight", ((2 if last else 6) * dim, dim), "adaln"), (q + "attn_norm_c.linear.bias", ((2 if last else 6) * dim,), "adaln_bias"),
              (q + "attn_norm_x.linear.weight", (6 * dim, dim), "adaln"), (q + "attn_norm_x.linear.bias", (6 * dim,), "adaln_bias")]
        for nm in ("to_q", "to_k", "to_v", "to_k_c", "to_v_c", "to_q_c"):
            s += [(f"{q}attn.{nm}.weight", (inner, dim), "linear"), (f"{q}attn.{nm}.bias", (inner,), "bias")]
        s += [(q + "attn.to_out.0.weight", (dim, inner), "linear"), (q + "attn.to_out.0.bias", (dim,), "bias")]
        if not last:
            s += [(q + "attn.to_out_c.weight", (dim, inner), "linear"), (q + "attn.to_out_c.bias", (dim,), "bias")]
            s += [(q + "ff_c.ff.0.0.weight", (dim * ff_mult, dim), "linear"), (q + "ff_c.ff.0.0.bias", (dim * ff_mult,), "bias"),
                  (q + "ff_c.ff.2.weight", (dim, dim * ff_mult), "linear"), (q + "ff_c.ff.2.bias", (dim,), "bias")]
        s += [(q + "ff_x.ff.0.0.weight", (dim * ff_mult, dim), "linear"), (q + "ff_x.ff.0.0.bias", (dim * ff_mult,), "bias"),
              (q + "ff_x.ff.2.weight", (dim, dim * ff_mult), "linear"), (q + "ff_x.ff.2.bias", (dim,), "bias")]
    s += [("transformer.norm_out.linear.weight", (2 * dim, dim), "adaln"), ("transformer.norm_out.linear.bias", (2 * dim,), "adaln_bias"),
          ("transformer.proj_out.weight", (mel_dim, dim), "linear"), ("transformer.proj_out.bias", (mel_dim,), "bias")]
    return s


def vocos_param_specs(in_ch=100, dim=512, inter=1536, layers=8, n_fft=1024):
    """Ordered (name, shape, kind) list for vocos 0.1.0 `charactr/vocos-mel-24khz`."""
    s = [("backbone.embed.weight", (dim, in_ch, 7), "conv"), ("backbone.embed.bias", (dim,), "bias"),
         ("backbone.norm.weight", (dim,), "gain"), ("backbone.norm.bias", (dim,), "bias")]
    for i in range(layers):
        q = f"backbone.convnext.{i}."
        s += [(q + "dwconv.weight", (dim, 1, 7), "conv"), (q + "dwconv.bias", (dim,), "bias"),
              (q + "norm.weight", (dim,), "gain"), (q + "norm.bias", (dim,), "bias"),
              (q + "pwconv1.weight", (inter, dim), "linear"), (q + "pwconv1.bias", (inter,), "bias"),
              (q + "pwconv2.weight", (dim, inter), "linear"), (q + "pwconv2.bias", (dim,), "bias"),
              (q + "gamma", (dim,), "layerscale")]
    s += [("backbone.final_layer_norm.weight", (dim,), "gain"), ("backbone.final_layer_norm.bias", (dim,), "bias"),
          ("head.out.weight", (n_fft + 2, dim), "head"), ("head.out.bias", (n_fft + 2,), "head_bias")]
    return s


def bigvgan_param_specs(num_mels=100, upsample_rates=(4, 4, 2, 2, 2, 2), upsample_kernel_sizes=(8, 8, 4, 4, 4, 4),
                        upsample_initial_channel=1536, resblock_kernel_sizes=(3, 7, 11), n_dil=3):
    """Ordered (name, shape, kind) list of the BigVGAN v2 generator after remove_weight_norm()."""
    c0 = upsample_initial_channel
    s = [("conv_pre.weight", (c0, num_mels, 7), "conv"), ("conv_pre.bias", (c0,), "bias")]
    for i, (r, k) in enumerate(zip(upsample_rates, upsample_kernel_sizes)):
        ci, co = c0 // 2 ** i, c0 // 2 ** (i + 1)
        s += [(f"ups.{i}.0.weight", (ci, co, k), "convT"), (f"ups.{i}.0.bias", (co,), "bias")]
    for i in range(len(upsample_rates)):
        ch = c0 // 2 ** (i + 1)
        for j, k in enumerate(resblock_kernel_sizes):
            q = f"resblocks.{i * len(resblock_kernel_sizes) + j}."
            for d in range(n_dil):
                s += [(f"{q}convs1.{d}.weight", (ch, ch, k), "conv_res"), (f"{q}convs1.{d}.bias", (ch,), "bias")]
            for d in range(n_dil):
                s += [(f"{q}convs2.{d}.weight", (ch, ch, k), "conv_res"), (f"{q}convs2.{d}.bias", (ch,), "bias")]
            for a in range(2 * n_dil):
                s += [(f"{q}activations.{a}.act.alpha", (ch,), "snake"), (f"{q}activations.{a}.act.beta", (ch,), "snake")]
    ch = c0 // 2 ** len(upsample_rates)
    s += [("activation_post.act.alpha", (ch,), "snake"), ("activation_post.act.beta", (ch,), "snake"),
          ("conv_post.weight", (1, ch, 7), "conv_post")]
    return s


def bigvgan_state_dict(seed=SEED_BIGVGAN, **arch):
    return make_state_dict(bigvgan_param_specs(**arch), seed)


def _draw(shape, kind, g):
    if kind in ("linear", "adaln"):
        return torch.randn(shape, generator=g) / math.sqrt(shape[1])
    if kind == "conv":
        return torch.randn(shape, generator=g) / math.sqrt(shape[1] * shape[2])
    if kind == "embed":
        return torch.randn(shape, generator=g)
    if kind == "convT":      # ConvTranspose1d weight [c_in, c_out, k]: each output sees c_in * k / stride taps
        return torch.randn(shape, generator=g) / math.sqrt(shape[0] * shape[2] / 2.0)
    if kind == "conv_res":   # residual-branch convs: half gain keeps the 18 residual adds per stage O(1)
        return torch.randn(shape, generator=g) * (0.5 / math.sqrt(shape[1] * shape[2]))
    if kind == "snake":      # log-scale alpha / beta around 0 (= 1 in linear scale)
        return torch.randn(shape, generator=g) * 0.2
    if kind == "conv_post":
        return torch.randn(shape, generator=g) * (0.2 / math.sqrt(shape[1] * shape[2]))
    if kind == "bias":
        return torch.randn(shape, generator=g) * 0.02
    if kind == "adaln_bias":  # non-zero so shift/scale/gate are O(0.3) and every branch is exercised
        return torch.randn(shape, generator=g) * 0.3
    if kind == "gain":
        return 1.0 + torch.randn(shape, generator=g) * 0.02
    if kind == "grn":  # zero-init in the reference (modules.py:228-229), which would hide GRN
        return torch.randn(shape, generator=g) * 0.1
    if kind == "layerscale":
        return 0.125 + torch.randn(shape, generator=g) * 0.01
    if kind == "head":  # keeps log-magnitudes ~N(0, 0.5^2) so exp() stays far from the 1e2 clip
        return torch.randn(shape, generator=g) * (0.5 / math.sqrt(shape[1]))
    if kind == "head_bias":
        return torch.randn(shape, generator=g) * 0.02
    raise ValueError(kind)


def make_state_dict(specs, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return {name: _draw(shape, kind, g).float().contiguous() for name, shape, kind in specs}


def dit_state_dict(seed=SEED_DIT, **arch):
    return make_state_dict(dit_param_specs(**arch), seed)


def unett_state_dict(seed=SEED_DIT, **arch):
    return make_state_dict(unett_param_specs(**arch), seed)


def mmdit_state_dict(seed=SEED_DIT, **arch):
    return make_state_dict(mmdit_param_specs(**arch), seed)


def vocos_state_dict(seed=SEED_VOCOS, **arch):
    return make_state_dict(vocos_param_specs(**arch), seed)


def ref_audio(n_samples=120_000, seed=SEED_AUDIO, amp=0.15):
    """0.15*sum_k sin(2 pi f_k n/24000 + phi_k)/sqrt(8) + 0.01*randn, rms ~ 0.11."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    f = 80.0 + (4000.0 - 80.0) * torch.rand(8, generator=g)
    ph = 2 * math.pi * torch.rand(8, generator=g)
    n = torch.arange(n_samples, dtype=torch.float64)
    w = torch.sin(2 * math.pi * f.double()[:, None] * n[None, :] / 24000.0 + ph.double()[:, None]).sum(0) / math.sqrt(8)
    w = amp * w.float() + 0.01 * torch.randn(n_samples, generator=g)
    return w[None, :].contiguous()


def text_ids(n_ref=60, n_gen=120, seed=SEED_TEXT, vocab=2545):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randint(1, vocab, (1, n_ref + n_gen), generator=g, dtype=torch.long)


def noise(n_frames, index=0, mel_dim=100):
    g = torch.Generator(device="cpu").manual_seed(SEED_NOISE + index)
    return torch.randn(n_frames, mel_dim, generator=g)
