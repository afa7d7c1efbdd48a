#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by EXECUTING THE REFERENCE'S OWN SOURCE FILES.

Runs only in the build container (needs /root/reference; the GPU box has none).
Nothing from the reference is copied: its modules are imported from where they
lie, fed seeded synthetic weights/inputs (tts-indic-server-f5_amd/synth.py) and
their outputs are stored as small .npz / .json vectors.

Third-party packages the reference imports but the container lacks
(torchdiffeq, x-transformers, torchaudio, librosa, jieba, pypinyin, vocos,
pydub) are replaced by leaf stand-ins taken from oracle/ (the same restated
functions the oracle uses), so those leaves are "parity unpinned" by
construction; everything in F/model/modules.py, backbones/dit.py,
backbones/unett.py, cfm.py and the pure-python glue of infer/utils_infer.py is
the reference's code running unmodified.

Usage:  python tests/golden/gen_golden.py            (rewrites tests/golden/*.npz, *.json)
"""
from __future__ import annotations

import importlib
import importlib.machinery
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/src/server/f5_tts"

from oracle import dit_oracle as O  # noqa: E402
from tts_indic_server_f5_amd import synth  # noqa: E402


def _mod(name, **kw):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__dict__.update(kw)
    sys.modules[name] = m
    return m


def install_leaf_shims():
    """Stand-ins for the absent third-party leaves."""
    _mod("torchdiffeq", odeint=lambda fn, y0, t, **kw: O.euler_odeint(fn, y0, t))

    class RotaryEmbedding(torch.nn.Module):
        def __init__(self, dim):
            super().__init__()
            self.dim = dim

        def forward_from_seq_len(self, seq_len):
            return O.rotary_freqs(seq_len, self.dim), 1.0

    def apply_rotary_pos_emb(t, freqs, scale=1):
        assert scale == 1 or scale == 1.0
        return O.apply_rotary(t, freqs)

    class RMSNorm(torch.nn.Module):
        def __init__(self, dim):
            super().__init__()
            self.scale = dim ** 0.5
            self.g = torch.nn.Parameter(torch.ones(dim))

        def forward(self, x):
            return torch.nn.functional.normalize(x, dim=-1) * self.scale * self.g

    xt = _mod("x_transformers", RMSNorm=RMSNorm)
    xt.x_transformers = _mod("x_transformers.x_transformers", RotaryEmbedding=RotaryEmbedding,
                             apply_rotary_pos_emb=apply_rotary_pos_emb)
    ta = _mod("torchaudio")
    ta.transforms = _mod("torchaudio.transforms")
    lb = _mod("librosa")
    lb.filters = _mod("librosa.filters", mel=None)
    _mod("jieba", initialize=lambda: None, cut=lambda s: list(s))
    _mod("pypinyin", lazy_pinyin=None, Style=None)
    for n, p in [("f5_tts", REF), ("f5_tts.model", REF + "/model"), ("f5_tts.model.backbones", REF + "/model/backbones")]:
        m = _mod(n)
        m.__path__ = [p]


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {name}.npz  {os.path.getsize(path) / 1024:.0f} KB")


TINY = dict(dim=128, depth=2, heads=2, ff_mult=2, text_dim=64, conv_layers=2, text_num_embeds=40)


def build_ref_dit(dit_mod, arch, seed=synth.SEED_DIT):
    sd = synth.dit_state_dict(seed=seed, **arch)
    net = dit_mod.DiT(**{k: v for k, v in arch.items()}, mel_dim=100)
    names = [n for n, _ in net.named_parameters()]
    assert names == [k[len("transformer."):] for k in sd.keys()], "synth key order != reference named_parameters()"
    net.load_state_dict({k[len("transformer."):]: v for k, v in sd.items()}, strict=True)
    return net.eval(), sd


def main():
    torch.set_num_threads(8)
    install_leaf_shims()
    modules = importlib.import_module("f5_tts.model.modules")
    dit_mod = importlib.import_module("f5_tts.model.backbones.dit")
    cfm_mod = importlib.import_module("f5_tts.model.cfm")

    # ---- (0) parameter-order fixture -------------------------------------------------
    net, sd = build_ref_dit(dit_mod, TINY)
    base_names = [n for n, _ in dit_mod.DiT(dim=64, depth=22, heads=1, ff_mult=2, text_dim=16, conv_layers=4,
                                            text_num_embeds=8).named_parameters()]
    with open(os.path.join(HERE, "dit_param_order.json"), "w") as f:
        json.dump(base_names, f)

    # ---- (1) tiny DiT.forward, with and without mask, both CFG branches ----------------
    g = torch.Generator().manual_seed(11)
    b, n, nt = 3, 50, 21
    x = torch.randn(b, n, 100, generator=g)
    cond = torch.randn(b, n, 100, generator=g) * (torch.arange(n)[None, :, None] < 17)
    text = torch.randint(0, 40, (b, nt), generator=g)
    text[1, 15:] = -1
    text[2, 9:] = -1
    tm = torch.tensor(0.37)
    lens = torch.tensor([50, 41, 33])
    mask = O.lens_to_mask(lens, n)
    outs = {}
    with torch.no_grad():
        for tag, da, dt in (("cond", False, False), ("null", True, True)):
            outs["out_b3_mask_" + tag] = net(x=x, cond=cond, text=text, time=tm, drop_audio_cond=da, drop_text=dt, mask=mask)
            outs["out_b1_" + tag] = net(x=x[:1], cond=cond[:1], text=text[:1], time=tm, drop_audio_cond=da, drop_text=dt, mask=None)
        outs["text_embed"] = net.text_embed(text[:1], n, drop_text=False)
        outs["text_embed_drop"] = net.text_embed(text[:1], n, drop_text=True)
        te = outs["text_embed"]
        outs["input_embed"] = net.input_embed(x[:1], cond[:1], te, drop_audio_cond=False)
        outs["time_embed"] = net.time_embed(tm.repeat(1))
        rope = net.rotary_embed.forward_from_seq_len(n)
        outs["block0"] = net.transformer_blocks[0](outs["input_embed"], outs["time_embed"], mask=None, rope=rope)
    save("dit_tiny_forward", x=x, cond=cond, text=text, time=tm, lens=lens, **outs)

    # ---- (2) tiny CFM.sample sweeps ---------------------------------------------------
    class NoMel(torch.nn.Identity):   # cond is always passed as mel [b, n, 100]; torchaudio is absent
        n_mel_channels = 100

    cfm = cfm_mod.CFM(transformer=net, mel_spec_module=NoMel(), num_channels=100,
                      odeint_kwargs=dict(method="euler")).eval()
    g = torch.Generator().manual_seed(12)
    cond1 = torch.randn(1, 17, 100, generator=g)
    text1 = torch.randint(0, 40, (1, 24), generator=g)
    cond3 = torch.randn(3, 17, 100, generator=g)
    text3 = torch.randint(0, 40, (3, 24), generator=g)
    text3[1, 20:] = -1
    text3[2, 11:] = -1
    cases = {}
    for steps in (4, 16):
        for sway in (None, -1.0):
            for cfg in (0.0, 2.0):
                key = f"s{steps}_sw{'n' if sway is None else 'm1'}_cfg{int(cfg)}"
                out, traj = cfm.sample(cond=cond1, text=text1, duration=48, steps=steps, cfg_strength=cfg,
                                       sway_sampling_coef=sway, seed=7)
                cases[key + "_b1_out"] = out
                cases[key + "_b1_traj_mid"] = traj[steps // 2]
    out, traj = cfm.sample(cond=cond3, text=text3, duration=torch.tensor([48, 40, 31]), steps=8, cfg_strength=2.0,
                           sway_sampling_coef=-1.0, seed=7)
    cases["b3_out"] = out
    cases["b3_traj1"] = traj[1]
    # text longer than cond -> lens extended; duration clamp lens+1
    out, _ = cfm.sample(cond=cond1[:, :10], text=text1, duration=12, steps=4, cfg_strength=2.0,
                        sway_sampling_coef=-1.0, seed=7)
    cases["longtext_out"] = out
    save("cfm_sample_tiny", cond1=cond1, text1=text1, cond3=cond3, text3=text3, **cases)

    # ---- (3) F5-Small real width, 3 blocks' worth of depth is not enough: full Small, n=160 ----
    small = dict(dim=768, depth=18, heads=12, ff_mult=2, text_dim=512, conv_layers=4, text_num_embeds=2545)
    net_s, _ = build_ref_dit(dit_mod, small)
    g = torch.Generator().manual_seed(13)
    n = 160
    xs = torch.randn(1, n, 100, generator=g)
    cs = torch.randn(1, n, 100, generator=g) * (torch.arange(n)[None, :, None] < 60)
    ts = torch.randint(1, 2545, (1, 40), generator=g)
    with torch.no_grad():
        o1 = net_s(x=xs, cond=cs, text=ts, time=torch.tensor(0.5), drop_audio_cond=False, drop_text=False)
        o2 = net_s(x=xs, cond=cs, text=ts, time=torch.tensor(0.5), drop_audio_cond=True, drop_text=True)
    save("dit_small_forward", x=xs, cond=cs, text=ts, out_cond=o1, out_null=o2)
    del net_s

    # ---- (4) full-size F5-Base forward at the C2 geometry: digest only -------------------
    base = dict(dim=1024, depth=22, heads=16, ff_mult=2, text_dim=512, conv_layers=4, text_num_embeds=2545)
    net_b, _ = build_ref_dit(dit_mod, base)
    n = 1404
    g = torch.Generator().manual_seed(14)
    xb = synth.noise(n, 0)[None]
    cb = torch.randn(1, n, 100, generator=g) * (torch.arange(n)[None, :, None] < 469)
    tb = synth.text_ids()
    with torch.no_grad():
        ob = net_b(x=xb, cond=cb, text=tb, time=torch.tensor(0.25), drop_audio_cond=False, drop_text=False)
    idx = torch.randperm(ob.numel(), generator=g)[:4096]
    save("dit_base_forward_digest", cond=cb.half(), idx=idx, sampled=ob.flatten()[idx],
         mean=ob.mean(), std=ob.std(), absmax=ob.abs().max())
    # ---- (4b) full CFM.sample at the C2 geometry (F5-Base, N = 1404, CFG 2, sway -1): digests at 4 and 32 NFE ----
    cfm_b = cfm_mod.CFM(transformer=net_b, mel_spec_module=NoMel(), num_channels=100, odeint_kwargs=dict(method="euler")).eval()
    gc = torch.Generator().manual_seed(14)
    cond_b = torch.randn(1, 469, 100, generator=gc)
    for steps in (4, 32):
        out, _ = cfm_b.sample(cond=cond_b, text=tb, duration=1404, steps=steps, cfg_strength=2.0, sway_sampling_coef=-1.0,
                              seed=synth.SEED_NOISE)
        gen = out[0, 469:]
        idx = torch.randperm(gen.numel(), generator=gc)[:16384]
        save(f"cfm_base_sample_digest_s{steps}", idx=idx, sampled=gen.flatten()[idx], mean=gen.mean(), std=gen.std(),
             absmax=gen.abs().max(), cond_head=out[0, :4])
        print("base sample", steps, "done", flush=True)
    del net_b, cfm_b

    # ---- (4c) UNetT (E2-TTS): parameter order, tiny forward (b=1, both CFG branches), tiny CFM.sample, Small forward ----
    gen_unett_fixtures(cfm_mod, NoMel)
    gen_e2base_fixtures(cfm_mod, NoMel)
    gen_mmdit_fixtures(cfm_mod, NoMel)

    # ---- (5) chunk_text / glue: reference's pure-python functions -----------------------
    gen_glue_fixtures()


UTINY = dict(dim=128, depth=4, heads=2, ff_mult=4, text_num_embeds=40)


def gen_unett_fixtures(cfm_mod, NoMel):
    unett_mod = importlib.import_module("f5_tts.model.backbones.unett")

    def build(arch):
        sd = synth.unett_state_dict(**arch)
        net = unett_mod.UNetT(**arch, mel_dim=100)
        names = [n for n, _ in net.named_parameters()]
        assert names == [k[len("transformer."):] for k in sd.keys()], "synth UNetT key order != reference named_parameters()"
        net.load_state_dict({k[len("transformer."):]: v for k, v in sd.items()}, strict=True)
        return net.eval()

    net = build(UTINY)
    with open(os.path.join(HERE, "unett_param_order.json"), "w") as f:
        json.dump([n for n, _ in net.named_parameters()], f)
    g = torch.Generator().manual_seed(31)
    n = 45
    x = torch.randn(1, n, 100, generator=g)
    cond = torch.randn(1, n, 100, generator=g) * (torch.arange(n)[None, :, None] < 15)
    text = torch.randint(0, 40, (1, 19), generator=g)
    tm = torch.tensor(0.41)
    outs = {}
    with torch.no_grad():
        outs["out_cond"] = net(x=x, cond=cond, text=text, time=tm, drop_audio_cond=False, drop_text=False)
        outs["out_null"] = net(x=x, cond=cond, text=text, time=tm, drop_audio_cond=True, drop_text=True)
    cfm = cfm_mod.CFM(transformer=net, mel_spec_module=NoMel(), num_channels=100, odeint_kwargs=dict(method="euler")).eval()
    out, traj = cfm.sample(cond=cond[:, :15], text=text, duration=45, steps=8, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=9)
    save("unett_tiny", x=x, cond=cond, text=text, time=tm, sample_out=out, sample_traj1=traj[1], **outs)
    small = dict(dim=768, depth=20, heads=12, ff_mult=4, text_num_embeds=2545)
    net_s = build(small)
    g = torch.Generator().manual_seed(32)
    n = 150
    xs = torch.randn(1, n, 100, generator=g)
    cs = torch.randn(1, n, 100, generator=g) * (torch.arange(n)[None, :, None] < 50)
    ts = torch.randint(1, 2545, (1, 40), generator=g)
    with torch.no_grad():
        o1 = net_s(x=xs, cond=cs, text=ts, time=torch.tensor(0.5), drop_audio_cond=False, drop_text=False)
    save("unett_small_forward", x=xs, cond=cs, text=ts, out_cond=o1)


def gen_e2base_fixtures(cfm_mod, NoMel, nfe_list=(8, 64)):
    """BASELINE configs[4] geometry: E2-TTS Base (UNetT 1024/24/16, ff x4), one 20 s chunk behind a 5 s reference = 2340 frames,
    60 + 240 token ids, CFG 2, sway -1: digests of the reference's own UNetT.forward and CFM.sample (8 and 64 NFE)."""
    unett_mod = importlib.import_module("f5_tts.model.backbones.unett")
    arch = dict(dim=1024, depth=24, heads=16, ff_mult=4, text_num_embeds=2545)
    sd = synth.unett_state_dict(**arch)
    net = unett_mod.UNetT(**arch, mel_dim=100)
    net.load_state_dict({k[len("transformer."):]: v for k, v in sd.items()}, strict=True)
    net = net.eval()
    n, n_ref = 2340, 469
    g = torch.Generator().manual_seed(51)
    xb = synth.noise(n, 0)[None]
    cb = torch.randn(1, n, 100, generator=g) * (torch.arange(n)[None, :, None] < n_ref)
    tb = synth.text_ids(60, 240)
    with torch.no_grad():
        ob = net(x=xb, cond=cb, text=tb, time=torch.tensor(0.25), drop_audio_cond=False, drop_text=False)
    idx = torch.randperm(ob.numel(), generator=g)[:4096]
    save("unett_base_forward_digest", cond=cb.half(), idx=idx, sampled=ob.flatten()[idx], mean=ob.mean(), std=ob.std(),
         absmax=ob.abs().max())
    print("e2-base forward done", flush=True)
    cfm = cfm_mod.CFM(transformer=net, mel_spec_module=NoMel(), num_channels=100, odeint_kwargs=dict(method="euler")).eval()
    gc = torch.Generator().manual_seed(52)
    cond = torch.randn(1, n_ref, 100, generator=gc)
    for steps in nfe_list:
        out, _ = cfm.sample(cond=cond, text=tb, duration=n, steps=steps, cfg_strength=2.0, sway_sampling_coef=-1.0,
                            seed=synth.SEED_NOISE)
        gen = out[0, n_ref:]
        idx = torch.randperm(gen.numel(), generator=gc)[:16384]
        save(f"cfm_e2base_sample_digest_s{steps}", idx=idx, sampled=gen.flatten()[idx], mean=gen.mean(), std=gen.std(),
             absmax=gen.abs().max(), cond_head=out[0, :4])
        print("e2-base sample", steps, "done", flush=True)


MMTINY = dict(dim=128, depth=3, heads=2, ff_mult=2, text_num_embeds=40)


def gen_mmdit_fixtures(cfm_mod, NoMel):
    """The reference's own MMDiT (F/model/backbones/mmdit.py) on seeded synthetic weights: forward with both CFG branches at b = 1 and at
    a padded b = 3 with mask (the joint-attention mask covers the audio keys only), the first block's two streams, and CFM.sample over
    it (8 steps, CFG 2, sway -1).  depth 3 = two full dual-stream blocks + the context-pre-only last block."""
    mm = importlib.import_module("f5_tts.model.backbones.mmdit")
    sd = synth.mmdit_state_dict(**MMTINY)
    net = mm.MMDiT(**MMTINY, mel_dim=100)
    names = [n for n, _ in net.named_parameters()]
    assert names == [k[len("transformer."):] for k in sd.keys()], "synth key order != reference MMDiT.named_parameters()"
    net.load_state_dict({k[len("transformer."):]: v for k, v in sd.items()}, strict=True)
    net = net.eval()
    g = torch.Generator().manual_seed(31)
    b, n, nt = 3, 50, 21
    x = torch.randn(b, n, 100, generator=g)
    cond = torch.randn(b, n, 100, generator=g) * (torch.arange(n)[None, :, None] < 17)
    text = torch.randint(0, 40, (b, nt), generator=g)
    text[1, 15:] = -1
    text[2, 9:] = -1
    tm = torch.tensor(0.37)
    lens = torch.tensor([50, 41, 33])
    mask = O.lens_to_mask(lens, n)
    outs = {}
    with torch.no_grad():
        for tag, da, dt in (("cond", False, False), ("null", True, True)):
            outs["out_b3_mask_" + tag] = net(x=x, cond=cond, text=text, time=tm, drop_audio_cond=da, drop_text=dt, mask=mask)
            outs["out_b1_" + tag] = net(x=x[:1], cond=cond[:1], text=text[:1], time=tm, drop_audio_cond=da, drop_text=dt, mask=None)
        t = net.time_embed(tm.repeat(1))
        c0 = net.text_embed(text[:1], drop_text=False)
        x0 = net.audio_embed(x[:1], cond[:1], drop_audio_cond=False)
        rope, c_rope = net.rotary_embed.forward_from_seq_len(n), net.rotary_embed.forward_from_seq_len(nt)
        c1, x1 = net.transformer_blocks[0](x0, c0, t, mask=None, rope=rope, c_rope=c_rope)
        outs.update(text_embed=c0, audio_embed=x0, block0_c=c1, block0_x=x1)
    cfm = cfm_mod.CFM(transformer=net, mel_spec_module=NoMel(), num_channels=100, odeint_kwargs=dict(method="euler")).eval()
    gs = torch.Generator().manual_seed(32)
    scond = torch.randn(1, 20, 100, generator=gs)
    stext = torch.randint(0, 40, (1, 14), generator=gs)
    sout, _ = cfm.sample(cond=scond, text=stext, duration=48, steps=8, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=7)
    save("mmdit_tiny", x=x, cond=cond, text=text, time=tm, lens=lens, sample_cond=scond, sample_text=stext, sample_out=sout, **outs)
    with open(os.path.join(HERE, "mmdit_param_order.json"), "w") as f:
        json.dump(names, f)


def gen_glue_fixtures():
    """infer/utils_infer.py needs torchaudio/pydub/vocos/matplotlib at import; give it empty stand-ins and use
    only its pure-python functions (chunk_text) and infer_batch_process's host glue with stub model/vocoder."""
    for name in ("pydub", "vocos", "vocos.feature_extractors"):
        _mod(name)
    _mod("transformers", pipeline=None)   # only used by the reference's ASR helper (out of scope)
    sys.modules["pydub"].AudioSegment = None
    sys.modules["pydub"].silence = None
    sys.modules["vocos"].Vocos = None
    fm = sys.modules["f5_tts.model"]
    fm.CFM = importlib.import_module("f5_tts.model.cfm").CFM
    ui = importlib.import_module("f5_tts.infer.utils_infer")

    story = open(REF + "/infer/examples/multi/story.txt", encoding="utf-8").read()
    basic = ("I don't really care what you call me. I've been a silent spectator, watching species evolve, "
             "empires rise and fall. But always remember, I am mighty and enduring.")
    texts = [basic, story, "", "no punctuation at all just words " * 10, "短句。第二句，很长的一句话；还有！",
             "ಕನ್ನಡ ಪಠ್ಯ. ಇನ್ನೊಂದು ವಾಕ್ಯ, ಮತ್ತೆ ಒಂದು! ಕೊನೆಯದು?", "a.b,c;d:e!f?g", "trailing space.   next one.  "]
    cases = []
    for t in texts:
        for mc in (20, 60, 135, 400):
            cases.append(dict(text=t, max_chars=mc, chunks=ui.chunk_text(t, max_chars=mc)))
    with open(os.path.join(HERE, "chunk_text.json"), "w", encoding="utf-8") as f:
        json.dump(cases, f, ensure_ascii=False, indent=0)
    print("wrote chunk_text.json", len(cases), "cases")

    # infer_batch_process host glue: stub sampler/vocoder with closed-form outputs so only the reference's own
    # rms / duration / strip / cross-fade arithmetic is captured.
    calls = []

    class StubModel:
        def sample(self, cond, text, duration, steps, cfg_strength, sway_sampling_coef):
            calls.append(dict(nw=int(cond.shape[-1]), text="".join(text[0]), duration=int(duration), steps=steps))
            n = duration
            base = torch.arange(n * 100, dtype=torch.float32).reshape(1, n, 100) / (n * 100)
            return base * (1 + len(calls)), None

    class StubVocoder:
        def decode(self, mel):
            t = mel.shape[-1]
            k = torch.arange(256 * (t - 1), dtype=torch.float32)
            return (torch.sin(k * 0.01) * mel.mean())[None]

    ui.convert_char_to_pinyin = lambda lst: [list(s) for s in lst]   # ASCII text: char passthrough
    glue = []
    for amp, sr, ch in ((0.3, 24000, 1), (0.02, 24000, 2)):
        calls.clear()
        g = torch.Generator().manual_seed(21)
        audio = torch.randn(ch, 24000 * 2 + 77, generator=g) * amp
        ref_text = "Some call me nature, others call me mother nature. "
        gens = ["I do not care.", "I have been a silent spectator, watching.", "Short."]
        wave, osr, spec = ui.infer_batch_process((audio, sr), ref_text, gens, StubModel(), StubVocoder(),
                                                 nfe_step=4, device="cpu")
        glue.append(dict(amp=amp, sr=sr, ch=ch, calls=list(calls), wave_dtype=str(wave.dtype), n=len(wave),
                         spec_shape=list(spec.shape)))
        save(f"glue_case_amp{amp}", audio=audio, wave=wave, spec=spec)
    with open(os.path.join(HERE, "glue_calls.json"), "w") as f:
        json.dump(glue, f, indent=0)
    print("wrote glue fixtures")


if __name__ == "__main__":
    if "--glue-only" in sys.argv:
        install_leaf_shims()
        gen_glue_fixtures()
    elif "--unett-only" in sys.argv:
        install_leaf_shims()
        importlib.import_module("f5_tts.model.modules")

        class NoMel(torch.nn.Identity):
            n_mel_channels = 100

        gen_unett_fixtures(importlib.import_module("f5_tts.model.cfm"), NoMel)
    elif "--mmdit-only" in sys.argv:
        install_leaf_shims()
        importlib.import_module("f5_tts.model.modules")

        class NoMel(torch.nn.Identity):
            n_mel_channels = 100

        gen_mmdit_fixtures(importlib.import_module("f5_tts.model.cfm"), NoMel)
    elif "--e2base-only" in sys.argv:
        torch.set_num_threads(8)
        install_leaf_shims()
        importlib.import_module("f5_tts.model.modules")

        class NoMel(torch.nn.Identity):
            n_mel_channels = 100

        gen_e2base_fixtures(importlib.import_module("f5_tts.model.cfm"), NoMel)
    else:
        main()
