"""CPU: the oracle restatement reproduces the fixtures produced by executing the reference's own
modules.py / dit.py / cfm.py (tests/golden/gen_golden.py)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import dit_oracle as O
from tts_indic_server_f5_amd import synth

TINY = dict(dim=128, depth=2, heads=2, ff_mult=2, text_dim=64, conv_layers=2, text_num_embeds=40)
TINY_CFG = O.DiTConfig(**TINY)


def _load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def _close(a, b, atol=2e-5, rtol=1e-5):
    assert a.shape == b.shape
    err = (a - b).abs().max().item()
    assert torch.allclose(a, b, atol=atol, rtol=rtol), f"max abs err {err}"


def test_param_order_matches_reference(golden_dir):
    names = json.load(open(os.path.join(golden_dir, "dit_param_order.json")))
    mine = [n[len("transformer."):] for n, _, _ in synth.dit_param_specs(dim=64, depth=22, heads=1, text_dim=16,
                                                                         conv_layers=4, text_num_embeds=8)]
    assert mine == names


def test_param_count_base():
    n = sum(int(np.prod(s)) for _, s, _ in synth.dit_param_specs())
    assert n == 337_096_804  # F5-Base with the shipped 2545-token vocab


def test_tiny_forward_submodules(golden_dir):
    g = _load(golden_dir, "dit_tiny_forward")
    sd = synth.dit_state_dict(**TINY)
    x, cond, text, tm = g["x"], g["cond"], g["text"], g["time"]
    n = x.shape[1]
    _close(O.time_embed(sd, tm.repeat(1)), g["time_embed"])
    te = O.text_embed(sd, TINY_CFG, text[:1], n, False)
    _close(te, g["text_embed"])
    _close(O.text_embed(sd, TINY_CFG, text[:1], n, True), g["text_embed_drop"])
    ie = O.input_embed(sd, x[:1], cond[:1], te, False)
    _close(ie, g["input_embed"])
    blk = O.dit_block(sd, "transformer.transformer_blocks.0.", TINY_CFG, ie, g["time_embed"], None,
                      O.rotary_freqs(n, 64))
    _close(blk, g["block0"], atol=5e-5)


@pytest.mark.parametrize("tag,da,dt", [("cond", False, False), ("null", True, True)])
def test_tiny_forward(golden_dir, tag, da, dt):
    g = _load(golden_dir, "dit_tiny_forward")
    sd = synth.dit_state_dict(**TINY)
    x, cond, text, tm = g["x"], g["cond"], g["text"], g["time"]
    mask = O.lens_to_mask(g["lens"], x.shape[1])
    out = O.dit_forward(sd, TINY_CFG, x, cond, text, tm, da, dt, mask)
    _close(out, g["out_b3_mask_" + tag], atol=1e-4)
    out1 = O.dit_forward(sd, TINY_CFG, x[:1], cond[:1], text[:1], tm, da, dt, None)
    _close(out1, g["out_b1_" + tag], atol=1e-4)


def test_cfm_sample_sweep(golden_dir):
    g = _load(golden_dir, "cfm_sample_tiny")
    sd = synth.dit_state_dict(**TINY)
    for steps in (4, 16):
        for sway in (None, -1.0):
            for cfg in (0.0, 2.0):
                key = f"s{steps}_sw{'n' if sway is None else 'm1'}_cfg{int(cfg)}"
                out, traj = O.cfm_sample(sd, TINY_CFG, g["cond1"], g["text1"], 48, steps=steps, cfg_strength=cfg,
                                         sway_sampling_coef=sway, seed=7)
                _close(out, g[key + "_b1_out"], atol=3e-4, rtol=1e-4)
                _close(traj[steps // 2], g[key + "_b1_traj_mid"], atol=3e-4, rtol=1e-4)
    out, traj = O.cfm_sample(sd, TINY_CFG, g["cond3"], g["text3"], torch.tensor([48, 40, 31]), steps=8,
                             cfg_strength=2.0, sway_sampling_coef=-1.0, seed=7)
    _close(out, g["b3_out"], atol=3e-4, rtol=1e-4)
    _close(traj[1], g["b3_traj1"], atol=3e-4, rtol=1e-4)
    out, _ = O.cfm_sample(sd, TINY_CFG, g["cond1"][:, :10], g["text1"], 12, steps=4, cfg_strength=2.0,
                          sway_sampling_coef=-1.0, seed=7)
    _close(out, g["longtext_out"], atol=3e-4, rtol=1e-4)


def test_small_forward(golden_dir):
    g = _load(golden_dir, "dit_small_forward")
    sd = synth.dit_state_dict(dim=768, depth=18, heads=12)
    tm = torch.tensor(0.5)
    o1 = O.dit_forward(sd, O.F5_SMALL, g["x"], g["cond"], g["text"], tm, False, False)
    o2 = O.dit_forward(sd, O.F5_SMALL, g["x"], g["cond"], g["text"], tm, True, True)
    _close(o1, g["out_cond"], atol=5e-4, rtol=1e-4)
    _close(o2, g["out_null"], atol=5e-4, rtol=1e-4)


# ---- known-answer tests for the third-party leaves the reference does not pin ----

def test_euler_closed_form():
    a = -0.7
    t = O.sway_time_grid(16, -1.0)
    y = O.euler_odeint(lambda tt, yy: a * yy, torch.ones(3), t, keep_trajectory=False)
    expect = torch.prod(1 + a * (t[1:] - t[:-1]))
    assert torch.allclose(y, expect.expand(3), atol=1e-6)


def test_midpoint_closed_form():
    """dy/dt = a y: one explicit-midpoint step multiplies y by 1 + a dt + (a dt)^2 / 2; second order, so closer to exp(a) than Euler."""
    a = -0.7
    t = O.sway_time_grid(16, -1.0)
    y = O.midpoint_odeint(lambda tt, yy: a * yy, torch.ones(3), t, keep_trajectory=False)
    dt = t[1:] - t[:-1]
    expect = torch.prod(1 + a * dt + 0.5 * (a * dt) ** 2)
    assert torch.allclose(y, expect.expand(3), atol=1e-6)
    e = O.euler_odeint(lambda tt, yy: a * yy, torch.ones(3), t, keep_trajectory=False)
    exact = torch.exp(torch.tensor(a))
    assert (y[0] - exact).abs() < 0.1 * (e[0] - exact).abs()
    # time-dependent field dy/dt = 2 t: the midpoint rule integrates it exactly on any grid
    y2 = O.midpoint_odeint(lambda tt, yy: 2 * tt * torch.ones_like(yy), torch.zeros(2), t, keep_trajectory=False)
    assert torch.allclose(y2, torch.ones(2), atol=1e-6)


def test_sway_grid_endpoints_monotone():
    for s in (None, -1.0, 0.5):
        t = O.sway_time_grid(32, s)
        assert abs(t[0].item()) < 1e-7 and abs(t[-1].item() - 1) < 1e-6
        assert (t[1:] > t[:-1]).all()
    t = O.sway_time_grid(32, -1.0)
    assert torch.allclose(t, 1 - torch.cos(torch.pi / 2 * torch.linspace(0, 1, 33)), atol=1e-6)


def test_rotary_properties():
    n = 37
    fr = O.rotary_freqs(n, 64)
    assert fr.shape == (1, n, 64)
    assert torch.equal(fr[0, :, 0::2], fr[0, :, 1::2])            # interleaved pairs
    x = torch.randn(2, n, 128)
    y = O.apply_rotary(x, fr)
    assert torch.equal(y[..., 64:], x[..., 64:])                    # heads >= 1 untouched
    assert torch.allclose(y[:, 0, :64], x[:, 0, :64])               # position 0 identity
    assert torch.allclose(y[..., :64].norm(dim=-1), x[..., :64].norm(dim=-1), atol=1e-4)
    # relative-position property on head 0: <R_m q, R_n k> depends on m-n only
    q, k = torch.randn(64), torch.randn(64)
    Q = O.apply_rotary(q.expand(1, n, 64).clone(), fr)[0]
    K = O.apply_rotary(k.expand(1, n, 64).clone(), fr)[0]
    s = Q @ K.T
    assert torch.allclose(s[5, 2], s[20, 17], atol=1e-3) and torch.allclose(s[9, 30], s[0, 21], atol=1e-3)


def test_mask_is_key_padding_only():
    sd = synth.dit_state_dict(**TINY)
    x = torch.randn(2, 12, 128)
    mask = O.lens_to_mask(torch.tensor([12, 7]), 12)
    o = O.attention(sd, "transformer.transformer_blocks.0.attn.", TINY_CFG, x, mask, None)
    assert (o[1, 7:] == 0).all() and (o[1, :7] != 0).any()
    # row 1 on its own 7 frames, no mask, must equal the masked batch result
    o1 = O.attention(sd, "transformer.transformer_blocks.0.attn.", TINY_CFG, x[1:, :7], None, None)
    assert torch.allclose(o[1, :7], o1[0], atol=1e-5)


def test_unett_oracle_vs_reference_fixture(golden_dir):
    """UNetT (E2-TTS) restatement vs the reference's own unett.py (fixture), incl. CFM.sample through it."""
    names = json.load(open(os.path.join(golden_dir, "unett_param_order.json")))
    ut = dict(dim=128, depth=4, heads=2, ff_mult=4, text_num_embeds=40)
    assert [n[len("transformer."):] for n, _, _ in synth.unett_param_specs(**ut)] == names
    g = _load(golden_dir, "unett_tiny")
    sd = synth.unett_state_dict(**ut)
    cfg = O.UNetTConfig(**ut)
    _close(O.unett_forward(sd, cfg, g["x"], g["cond"], g["text"], g["time"], False, False), g["out_cond"], atol=1e-4)
    _close(O.unett_forward(sd, cfg, g["x"], g["cond"], g["text"], g["time"], True, True), g["out_null"], atol=1e-4)
    out, traj = O.cfm_sample(sd, cfg, g["cond"][:, :15], g["text"], 45, steps=8, cfg_strength=2.0, sway_sampling_coef=-1.0,
                             seed=9, forward_fn=lambda **kw: O.unett_forward(sd, cfg, **kw))
    _close(out, g["sample_out"], atol=3e-4, rtol=1e-4)
    _close(traj[1], g["sample_traj1"], atol=3e-4, rtol=1e-4)


def test_rms_norm_scale():
    x = torch.randn(3, 5, 64)
    y = O.rms_norm(x, torch.ones(64))
    assert torch.allclose(y.pow(2).mean(-1), torch.ones(3, 5), atol=1e-5)   # unit RMS: ||y|| = sqrt(D)


# ---- MMDiT (F/model/backbones/mmdit.py): the oracle vs the reference's own forward / CFM.sample on seeded weights ----

MMTINY = dict(dim=128, depth=3, heads=2, ff_mult=2, text_num_embeds=40)


def test_mmdit_param_order_matches_reference(golden_dir):
    names = json.load(open(os.path.join(golden_dir, "mmdit_param_order.json")))
    assert names == [k[len("transformer."):] for k in synth.mmdit_state_dict(**MMTINY).keys()]


def test_mmdit_oracle_vs_reference_fixture(golden_dir):
    g = _load(golden_dir, "mmdit_tiny")
    sd, cfg = synth.mmdit_state_dict(**MMTINY), O.MMDiTConfig(**MMTINY)
    mask = O.lens_to_mask(g["lens"], g["x"].shape[1])
    for tag, da, dt in (("cond", False, False), ("null", True, True)):
        o3 = O.mmdit_forward(sd, cfg, g["x"], g["cond"], g["text"], g["time"], da, dt, mask=mask)
        _close(o3, g["out_b3_mask_" + tag], atol=3e-4, rtol=1e-4)
        o1 = O.mmdit_forward(sd, cfg, g["x"][:1], g["cond"][:1], g["text"][:1], g["time"], da, dt)
        _close(o1, g["out_b1_" + tag], atol=3e-4, rtol=1e-4)
    c0 = O.mmdit_text_embed(sd, cfg, g["text"][:1], False)
    x0 = O.mmdit_audio_embed(sd, g["x"][:1], g["cond"][:1], False)
    _close(c0, g["text_embed"], atol=1e-5, rtol=1e-5)
    _close(x0, g["audio_embed"], atol=1e-4, rtol=1e-4)
    t = O.time_embed(sd, g["time"].repeat(1))
    n, nt = g["x"].shape[1], g["text"].shape[1]
    c1, x1 = O.mmdit_block(sd, "transformer.transformer_blocks.0.", cfg, x0, c0, t, None, O.rotary_freqs(n), O.rotary_freqs(nt), False)
    _close(c1, g["block0_c"], atol=2e-4, rtol=1e-4)
    _close(x1, g["block0_x"], atol=2e-4, rtol=1e-4)
    out, _ = O.cfm_sample(sd, cfg, g["sample_cond"], g["sample_text"], 48, steps=8, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=7,
                          forward_fn=lambda **kw: O.mmdit_forward(sd, cfg, **kw), keep_trajectory=False)
    _close(out, g["sample_out"], atol=5e-4, rtol=1e-4)
