"""GPU parity at the workload sizes of BASELINE.json's configs (the round-1 verdict listed C1 / C3 / C4 / C5 as never exercised under a
check), through the C ABI:
  C1  F5-TTS-Small, 3 s utterance (N = 748), 16 NFE, CFG 2, Vocos              vs the CPU oracle
  C3  the per-GPU share of the 64-utterance batch: B = 8 (M = 22 528 rows, batch-mode GEMM tiles), identical and ragged units
  C4  BigVGAN v2 full geometry at 936 frames, B = 2 vs the CPU oracle, and B = 16 copies == single
  C5  E2-TTS Base (UNetT 1024 / 24 / 16, ff x4), one 20 s chunk = 2340 frames: forward and CFM.sample (8 and 64 NFE) against digests of
      the REFERENCE's own unett.py / cfm.py (tests/golden/gen_golden.py --e2base-only); 3-chunk infer_process with cross-fade
plus the reference's padded-batch sampler semantics (reference fixture b3_out), the fp16 outlier guard and an attention tile that
overhangs the workspace.  Tolerances: BASELINE.json north_star = 1e-3 RMS on mel frames, 1e-4 on waveform samples."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import bigvgan_oracle as B  # noqa: E402
from oracle import dit_oracle as O  # noqa: E402
from oracle import vocos_oracle as V  # noqa: E402
from tts_indic_server_f5_amd import infer, synth  # noqa: E402

TINY = dict(dim=128, depth=2, heads=2, ff_mult=2, text_dim=64, conv_layers=2, text_num_embeds=40)


def _load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def _report(tag, got, ref):
    got, ref = got.float().cpu(), ref.float().cpu()
    d = got - ref
    rms = d.pow(2).mean().sqrt().item()
    print(f"[parity] {tag}: rms_err {rms:.3e} max_err {d.abs().max():.3e} ref_rms {ref.pow(2).mean().sqrt():.3e}")
    return rms


def _counter(name):
    from tts_indic_server_f5_amd import _lib
    v = C.c_int64(0)
    _lib.check(_lib.lib().f5hip_get_counter(name.encode(), C.byref(v)), "get_counter")
    return v.value


def _reset_counters():
    from tts_indic_server_f5_amd import _lib
    _lib.check(_lib.lib().f5hip_get_counter(b"reset", None), "reset")


# ---------------------------------------------------------------------------------------------------------------- C1
def test_c1_small_16nfe_sample_and_vocos():
    from tts_indic_server_f5_amd.model import F5TTS_SMALL, F5HipModel
    from tts_indic_server_f5_amd.vocoder import F5HipVocos
    arch = dict(dim=768, depth=18, heads=12)
    sd, vsd = synth.dit_state_dict(**arch), synth.vocos_state_dict()
    n_ref, n = 468, 748                                    # 5 s reference + 3 s generated: 468 + int(468 / 60 * 36)
    g = torch.Generator().manual_seed(21)
    cond = torch.randn(1, n_ref + 1, 100, generator=g)
    text = synth.text_ids(60, 36)
    y0 = synth.noise(n, 0)[None]
    ref, _ = O.cfm_sample(sd, O.DiTConfig(**arch), cond, text, n, steps=16, cfg_strength=2.0, sway_sampling_coef=-1.0, y0=y0, keep_trajectory=False)
    m = F5HipModel(F5TTS_SMALL, sd)
    out, _ = m.sample(cond, text, n, steps=16, cfg_strength=2.0, sway_sampling_coef=-1.0, y0=y0)
    assert _report("C1 F5-Small 16 NFE mel", out[:, n_ref + 1:], ref[:, n_ref + 1:]) < 1e-3
    wave = F5HipVocos(vsd).decode(out[:, n_ref:].permute(0, 2, 1))
    wref = V.vocos_decode(vsd, ref[:, n_ref:].permute(0, 2, 1))
    err = (wave.cpu() - wref).abs().max().item()
    print(f"[parity] C1 waveform max err {err:.3e}")
    assert err < 1e-4 * max(1.0, wref.abs().max().item())


# ---------------------------------------------------------------------------------------------------------------- C3
@pytest.fixture(scope="module")
def base_model():
    from tts_indic_server_f5_amd.model import F5TTS_BASE, F5HipModel
    return F5HipModel(F5TTS_BASE, synth.dit_state_dict())


def test_c3_share_batch8_copies_equal_single(base_model, attn_shape_invariant):
    """8 identical utterances (M = 16 x 1408 rows: every block GEMM runs gemm6's 256 x 256 tiles) give, item by item, the batch-1 result
    (gemm5's 176 x 64 / 128 / 192 tiles in one round): only the kernel and its tile shapes differ, the k order of every dot product does not.
    (Shape-invariant attention arithmetic: see conftest.attn_shape_invariant; the default mode is compared in test_gpu_dit.)"""
    gc = torch.Generator().manual_seed(14)
    cond = torch.randn(1, 469, 100, generator=gc)
    text = synth.text_ids()
    y0 = synth.noise(1404, 0)[None]
    one, _ = base_model.sample(cond, text, 1404, steps=2, cfg_strength=2.0, sway_sampling_coef=-1.0, y0=y0)
    _reset_counters()
    eight, _ = base_model.sample(cond.expand(8, -1, -1), text.expand(8, -1), 1404, steps=2, cfg_strength=2.0, sway_sampling_coef=-1.0,
                                 y0=y0.expand(8, -1, -1))
    # 2 steps x 22 layers x 4 block GEMMs, all on the batch-mode kernel (gemm6: 256 x 256 tiles; the single utterance ran gemm5's 176-row tiles)
    assert _counter("gemm6") == 2 * 22 * 4 and _counter("gemm5_rb11") == 0
    for i in range(8):
        assert _report(f"C3 share item {i}", eight[i], one[0]) < 2e-5


def test_c3_ragged_batch8_is_per_item_batch1(base_model, attn_shape_invariant):
    """Ragged per-GPU share (generated lengths 6 .. 14 s): every item equals the item sampled alone."""
    gc = torch.Generator().manual_seed(15)
    cond = torch.randn(1, 469, 100, generator=gc)
    durs = [468 + d for d in (562, 700, 811, 936, 1000, 1111, 1250, 1312)]
    texts = torch.stack([synth.text_ids(seed=synth.SEED_TEXT + i)[0] for i in range(8)])
    y0 = [synth.noise(n, i) for i, n in enumerate(durs)]
    batch, _ = base_model.sample(cond.expand(8, -1, -1), texts, torch.tensor(durs), steps=2, cfg_strength=2.0, sway_sampling_coef=-1.0, y0=y0)
    for i in (0, 3, 7):
        single, _ = base_model.sample(cond, texts[i:i + 1], durs[i], steps=2, cfg_strength=2.0, sway_sampling_coef=-1.0, y0=[y0[i]])
        assert _report(f"C3 ragged item {i} (n={durs[i]})", batch[i, :durs[i]], single[0]) < 2e-5
        assert (batch[i, durs[i]:] == 0).all()


def test_c3_batch8_default_mode_vs_reference_digest(golden_dir, base_model):
    """C3's per-GPU share in the DEFAULT attention mode, tied to the reference directly: 8 copies of the C2 utterance, 32 NFE, every item
    against the digest of the reference's own CFM.sample (tests/golden/cfm_base_sample_digest_s32.npz) at north_star's 1e-3."""
    from tts_indic_server_f5_amd import _lib
    _lib.check(_lib.lib().f5hip_set_attention_shape_invariant(0), "set_attention_shape_invariant")
    g = _load(golden_dir, "cfm_base_sample_digest_s32")
    gc = torch.Generator().manual_seed(14)
    cond = torch.randn(1, 469, 100, generator=gc)
    out, _ = base_model.sample(cond.expand(8, -1, -1), synth.text_ids().expand(8, -1), 1404, steps=32, cfg_strength=2.0, sway_sampling_coef=-1.0,
                               seed=synth.SEED_NOISE)   # (the reference re-seeds per item, cfm.py:181-186: eight times the digest's noise)
    for i in range(8):
        got = out[i, 469:].cpu().flatten()[g["idx"]]
        assert _report(f"C3 default mode, item {i} of 8, 32 NFE vs reference digest", got, g["sampled"]) < 1e-3
        assert torch.equal(out[i, :4].cpu(), g["cond_head"])


# ---------------------------------------------------------------------------------------------------------------- C4
def test_c4_sampler_batch16_copies_equal_single(base_model, attn_shape_invariant):
    """C4's sampler half at its own batch (16 utterances: M = 32 x 1408 = 45 056 rows, 16 rounds of the batch-mode tiles): item == single."""
    gc = torch.Generator().manual_seed(14)
    cond = torch.randn(1, 469, 100, generator=gc)
    text = synth.text_ids()
    y0 = synth.noise(1404, 0)[None]
    one, _ = base_model.sample(cond, text, 1404, steps=2, cfg_strength=2.0, sway_sampling_coef=-1.0, y0=y0)
    many, _ = base_model.sample(cond.expand(16, -1, -1), text.expand(16, -1), 1404, steps=2, cfg_strength=2.0, sway_sampling_coef=-1.0,
                                y0=y0.expand(16, -1, -1))
    for i in (0, 7, 15):
        assert _report(f"C4 sampler item {i} of 16", many[i], one[0]) < 2e-5


def test_c4_bigvgan_full_geometry_936_frames():
    from tts_indic_server_f5_amd.vocoder import F5HipBigVGAN
    sd = synth.bigvgan_state_dict()
    voc = F5HipBigVGAN(sd)
    g = torch.Generator().manual_seed(78)
    mel = torch.randn(2, 100, 936, generator=g) * 1.5 - 1.0
    ref = B.bigvgan_forward(sd, B.BIGVGAN_V2_24K_100B_256X, mel)
    got = voc(mel)
    assert got.shape == ref.shape == (2, 1, 256 * 936)
    d = (got.cpu() - ref).abs()
    print(f"[parity] C4 BigVGAN T=936 B=2: max err {d.max():.3e} rms {d.pow(2).mean().sqrt():.3e} clipped {(ref.abs() >= 1).float().mean():.4f}")
    assert d.max().item() < 1e-4
    many = voc(mel[:1].expand(16, -1, -1))
    assert (many - got[:1]).abs().max().item() < 2e-6          # B = 16 copies == single (batch composition never leaks between items)
    # the fp16 fast mode (gemm_planes=3) at the same geometry: outside the 1e-4 parity bound by design, pinned at its measured level
    fast = F5HipBigVGAN(sd, gemm_planes=3)(mel)
    df = (fast.cpu() - ref).abs()
    print(f"[parity] C4 BigVGAN T=936 B=2 fp16 fast mode: max err {df.max():.3e} rms {df.pow(2).mean().sqrt():.3e}")
    assert df.max().item() < 3e-3 and df.pow(2).mean().sqrt().item() < 5e-4


# ---------------------------------------------------------------------------------------------------------------- C5
@pytest.fixture(scope="module", params=[3, 2], ids=["mixed_f16", "bf16x3"])
def e2base_model(request):
    """Both precision modes against the reference's own E2-Base digests: mixed (fp16 block GEMMs, the default) measures 4.9e-4 - 5.9e-4 rms,
    split bf16 everywhere 0.6e-4 - 1.5e-4, against the 1e-3 bound."""
    from tts_indic_server_f5_amd.model import E2TTS_BASE, F5HipModel
    return F5HipModel(E2TTS_BASE, synth.unett_state_dict(), gemm_planes=request.param)


def test_c5_e2base_forward_digest(golden_dir, e2base_model):
    g = _load(golden_dir, "unett_base_forward_digest")
    x = synth.noise(2340, 0)[None]
    out = e2base_model.transformer_forward(x, g["cond"].float(), synth.text_ids(60, 240), 0.25, False, False)
    got = out.flatten().cpu()[g["idx"]]
    assert _report("C5 E2-Base forward N=2340 (4096 sampled)", got, g["sampled"]) < 1e-3
    assert abs(out.mean().item() - float(g["mean"])) < 1e-3


@pytest.mark.parametrize("steps", [8, 64])
def test_c5_e2base_sample_vs_reference_digest(golden_dir, e2base_model, steps):
    g = _load(golden_dir, f"cfm_e2base_sample_digest_s{steps}")
    gc = torch.Generator().manual_seed(52)
    cond = torch.randn(1, 469, 100, generator=gc)
    out, _ = e2base_model.sample(cond, synth.text_ids(60, 240), 2340, steps=steps, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=synth.SEED_NOISE)
    gen = out[0, 469:].cpu()
    err = _report(f"C5 E2-Base sample {steps} NFE (16384 sampled generated-frame elements)", gen.flatten()[g["idx"]], g["sampled"])
    assert err < 1e-3
    assert torch.equal(out[0, :4].cpu(), g["cond_head"])


UARCH = dict(dim=256, depth=4, heads=4, ff_mult=4, text_num_embeds=96)
VOCAB = {chr(32 + i): i for i in range(96)}


def test_c5_three_chunk_infer_process_unett_batched_equals_sequential():
    """Long-form text on the E2 backbone: infer_process chunks it, the chunks are sampled in ONE batched call (sample_units) and
    cross-faded; the result must equal (a) the same HIP objects driven chunk by chunk like the reference does and (b) the CPU oracle
    pipeline."""
    from tts_indic_server_f5_amd.model import F5HipModel, UNetTArch
    from tts_indic_server_f5_amd.tokenizer import list_str_to_idx
    from tts_indic_server_f5_amd.vocoder import F5HipVocos
    sd, vsd = synth.unett_state_dict(**UARCH), synth.vocos_state_dict()
    ucfg = O.UNetTConfig(**UARCH)
    ref_audio = (synth.ref_audio(24000 * 2, amp=0.15), 24000)
    ref_text = "Some call me nature."
    gen_text = ("I do not care what you call me. I have been a silent spectator, watching species evolve. Always remember, I endure. "
                "Empires rise and fall, yet I remain. Your future depends on me. When I thrive, you thrive; when I falter, you falter. "
                "I have fed species greater than you, and I have starved species greater than you. My oceans, my soil, my flowing streams, "
                "my forests: they all can take you, or leave you. How you choose to live each day, whether you regard or disregard me, "
                "does not really matter to me. One way or the other, your actions will determine your fate, not mine.")
    kw = dict(nfe_step=8, cfg_strength=2.0, sway_sampling_coef=-1.0)
    chunks = infer.chunk_text(gen_text, max_chars=int(len(ref_text.encode()) / 2 * 23))
    assert len(chunks) >= 3
    hip, voc = F5HipModel(UNetTArch(**UARCH), sd, vocab_char_map=VOCAB), F5HipVocos(vsd)

    class Sequential:   # the reference's structure: one sample() call per chunk
        def sample(self, **k):
            return hip.sample(**k)

    class OracleModel:
        def sample(self, cond, text, duration, steps, cfg_strength, sway_sampling_coef):
            mel = V.vocos_mel_spectrogram(cond.cpu()).permute(0, 2, 1)
            out, _ = O.cfm_sample(sd, ucfg, mel, list_str_to_idx(text, VOCAB), duration, steps=steps, cfg_strength=cfg_strength,
                                  sway_sampling_coef=sway_sampling_coef, keep_trajectory=False,
                                  forward_fn=lambda **f: O.unett_forward(sd, ucfg, **f))
            return out, None

    class OracleVocoder:
        def decode(self, mel):
            return V.vocos_decode(vsd, mel.cpu())

    torch.manual_seed(5)
    w_b, sr, s_b = infer.infer_process(ref_audio, ref_text, gen_text, hip, voc, device="cuda", **kw)
    torch.manual_seed(5)
    w_s, _, s_s = infer.infer_process(ref_audio, ref_text, gen_text, Sequential(), voc, device="cuda", **kw)
    torch.manual_seed(5)
    w_r, _, s_r = infer.infer_process(ref_audio, ref_text, gen_text, OracleModel(), OracleVocoder(), **kw)
    assert w_b.dtype == np.float64 and w_b.shape == w_s.shape == w_r.shape            # float64 after the cross-fade
    print(f"[parity] C5 3-chunk: batched vs sequential wave {np.abs(w_b - w_s).max():.3e}, mel {np.sqrt(np.mean((s_b - s_s) ** 2)):.3e}; "
          f"vs oracle wave {np.abs(w_b - w_r).max():.3e}, mel {np.sqrt(np.mean((s_b - s_r) ** 2)):.3e}")
    assert np.abs(w_b - w_s).max() < 2e-5 and np.sqrt(np.mean((s_b - s_s) ** 2)) < 2e-5
    assert np.sqrt(np.mean((s_b - s_r) ** 2)) < 1e-3 and np.abs(w_b - w_r).max() < 1e-4


# ---------------------------------------------------------------------------------------------------------------- sampler modes
@pytest.fixture(scope="module", params=[2, 3], ids=["bf16x3", "mixed_f16"])
def tiny_model(request):
    from tts_indic_server_f5_amd.model import DiTArch, F5HipModel
    return F5HipModel(DiTArch(**TINY), synth.dit_state_dict(**TINY), gemm_planes=request.param)


def test_padded_batch_sampler_vs_reference_fixture(golden_dir, tiny_model):
    """What the reference computes when CFM.sample itself gets b = 3 unequal items (cfm.py:151-154 mask = lens_to_mask(duration);
    unmasked convolutions over the padding): fixture b3_out from the reference's own cfm.py, every row incl. the padded ones."""
    g = _load(golden_dir, "cfm_sample_tiny")
    out, _ = tiny_model.sample(g["cond3"], g["text3"], torch.tensor([48, 40, 31]), steps=8, cfg_strength=2.0, sway_sampling_coef=-1.0,
                               seed=7, padded_batch=True)
    assert out.shape == g["b3_out"].shape
    assert _report("padded-batch sample b=3 (all rows)", out, g["b3_out"]) < 1e-3
    assert torch.equal(out[:, :17].cpu(), g["b3_out"][:, :17])
    # and the default mode differs from it exactly where the reference's batch composition leaks (item 0 is the longest: identical)
    per_item, _ = tiny_model.sample(g["cond3"], g["text3"], torch.tensor([48, 40, 31]), steps=8, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=7)
    assert _report("padded vs batch-1 semantics, longest item", per_item[0], out[0]) < 1e-3


def test_lens_no_ref_audio_and_duplicate_test_branches(tiny_model):
    """Remaining knobs of CFM.sample (cfm.py:88-99): `lens` shorter than the prompt, `no_ref_audio`, `duplicate_test` -- vs the oracle."""
    sd, cfg = synth.dit_state_dict(**TINY), O.DiTConfig(**TINY)
    g = torch.Generator().manual_seed(61)
    cond = torch.randn(1, 24, 100, generator=g)
    text = torch.randint(0, 40, (1, 12), generator=g)
    y0 = torch.randn(1, 60, 100, generator=g)
    lens = torch.tensor([15])
    ref, _ = O.cfm_sample(sd, cfg, cond, text, 60, lens=lens, steps=4, cfg_strength=2.0, sway_sampling_coef=-1.0, y0=y0, keep_trajectory=False)
    got, _ = tiny_model.sample(cond, text, 60, lens=lens, steps=4, cfg_strength=2.0, sway_sampling_coef=-1.0, y0=y0)
    assert _report("lens < prompt", got, ref) < 1e-3 and torch.equal(got[:, :15].cpu(), ref[:, :15])
    ref, _ = O.cfm_sample(sd, cfg, cond, text, 60, steps=4, cfg_strength=2.0, sway_sampling_coef=-1.0, y0=y0, no_ref_audio=True, keep_trajectory=False)
    got, _ = tiny_model.sample(cond, text, 60, steps=4, cfg_strength=2.0, sway_sampling_coef=-1.0, y0=y0, no_ref_audio=True)
    assert _report("no_ref_audio", got, ref) < 1e-3 and (got[:, :24] == 0).all()
    # duplicate_test (cfm.py:139-141,190-194): y0 <- 0.9 y0 + 0.1 [0 | cond | 0...], t from 0.1, int(0.9 steps) steps: restated with the oracle's pieces
    t_inter, steps = 0.1, 10
    test_cond = torch.nn.functional.pad(cond, (0, 0, 24, 60 - 48))
    y0d = (1 - t_inter) * y0 + t_inter * test_cond
    s2 = int(steps * (1 - t_inter))
    tg = torch.linspace(t_inter, 1, s2 + 1)
    tg = tg + (-1.0) * (torch.cos(torch.pi / 2 * tg) - 1 + tg)
    step_cond = torch.nn.functional.pad(cond, (0, 0, 0, 36))
    fn = lambda t, x: (lambda p, q: p + (p - q) * 2.0)(O.dit_forward(sd, cfg, x=x, cond=step_cond, text=text, time=t, mask=None, drop_audio_cond=False, drop_text=False),
                                                       O.dit_forward(sd, cfg, x=x, cond=step_cond, text=text, time=t, mask=None, drop_audio_cond=True, drop_text=True))
    last = O.euler_odeint(fn, y0d, tg, keep_trajectory=False)
    ref = torch.cat([cond, last[:, 24:]], dim=1)
    got, _ = tiny_model.sample(cond, text, 60, steps=steps, cfg_strength=2.0, sway_sampling_coef=-1.0, y0=y0, duplicate_test=True, t_inter=t_inter)
    assert _report("duplicate_test", got, ref) < 1e-3


@pytest.mark.parametrize("planes", [2, 3], ids=["bf16x3", "mixed_f16"])
def test_midpoint_solver_vs_oracle(planes):
    """CFM(odeint_kwargs=dict(method="midpoint")) (cfm.py:37-41,200): two backbone evaluations per step, time embeddings at t_i and t_i + dt/2.
    torchdiffeq is absent, so the step rule itself is pinned by the closed-form test in tests/test_oracle_dit.py (parity unpinned leaf)."""
    from tts_indic_server_f5_amd._lib import F5HipError
    from tts_indic_server_f5_amd.model import DiTArch, F5HipModel
    sd, cfg = synth.dit_state_dict(**TINY), O.DiTConfig(**TINY)
    model = F5HipModel(DiTArch(**TINY), sd, gemm_planes=planes, odeint_kwargs=dict(method="midpoint"))
    g = torch.Generator().manual_seed(63)
    cond = torch.randn(1, 24, 100, generator=g)
    text = torch.randint(0, 40, (1, 12), generator=g)
    y0 = torch.randn(1, 60, 100, generator=g)
    kw = dict(steps=6, cfg_strength=2.0, sway_sampling_coef=-1.0, y0=y0)
    ref, _ = O.cfm_sample(sd, cfg, cond, text, 60, keep_trajectory=False, method="midpoint", **kw)
    eul, _ = O.cfm_sample(sd, cfg, cond, text, 60, keep_trajectory=False, **kw)
    got, _ = model.sample(cond, text, 60, **kw)
    assert _report("midpoint sample", got, ref) < 1e-3
    assert (ref - eul)[:, 24:].pow(2).mean().sqrt() > 1e-2          # a different trajectory from Euler, not a relabelled one
    with pytest.raises(F5HipError, match="midpoint"):
        model.sample(cond, text, 60, steps=65, cfg_strength=2.0, y0=y0)
    with pytest.raises(ValueError):
        F5HipModel(DiTArch(**TINY), sd, odeint_kwargs=dict(method="dopri5"))


def test_attention_tile_overhanging_the_workspace(tiny_model):
    """A last sequence whose final 256-query attention tile reaches past its 128-row padding (n % 256 in 1..128): those query rows are
    loaded from the slack rows of the workspace and never stored (the round-1 aborts came from exactly this read before the slack existed)."""
    sd, cfg = synth.dit_state_dict(**TINY), O.DiTConfig(**TINY)
    g = torch.Generator().manual_seed(71)
    for n in (300, 257, 384 + 1):
        x = torch.randn(1, n, 100, generator=g)
        cond = torch.randn(1, n, 100, generator=g) * (torch.arange(n)[None, :, None] < 40)
        text = torch.randint(0, 40, (1, 30), generator=g)
        ref = O.dit_forward(sd, cfg, x=x, cond=cond, text=text, time=torch.tensor(0.3), mask=None, drop_audio_cond=False, drop_text=False)
        got = tiny_model.transformer_forward(x, cond, text, 0.3, False, False)
        assert _report(f"overhanging tile n={n}", got, ref) < 1e-3


def test_fp16_outlier_channel_does_not_poison_the_row():
    """Mixed mode writes the FF1 / GELU output as ONE fp16 plane.  A feed-forward channel whose activation exceeds the fp16 range (65 504)
    but whose FF2 weights are zero must leave the output untouched: with an unsaturated convert it becomes inf, inf x 0 = NaN in the fp32
    accumulator, and the whole row is lost (the reference runs these activations in fp32: F/infer/utils_infer.py:176-184)."""
    from tts_indic_server_f5_amd.model import DiTArch, F5HipModel
    sd = {k: v.clone() for k, v in synth.dit_state_dict(**TINY).items()}
    ch = 5
    sd["transformer.transformer_blocks.0.ff.ff.0.0.weight"][ch] *= 3.0e5      # pre-GELU activation of channel 5: O(1e5)
    sd["transformer.transformer_blocks.0.ff.ff.2.weight"][:, ch] = 0.0        # ... which the second projection ignores
    g = torch.Generator().manual_seed(81)
    x = torch.randn(1, 200, 100, generator=g)
    cond = torch.randn(1, 200, 100, generator=g) * (torch.arange(200)[None, :, None] < 50)
    text = torch.randint(0, 40, (1, 30), generator=g)
    ref = O.dit_forward(sd, O.DiTConfig(**TINY), x=x, cond=cond, text=text, time=torch.tensor(0.4), mask=None, drop_audio_cond=False, drop_text=False)
    strict = F5HipModel(DiTArch(**TINY), sd, gemm_planes=2).transformer_forward(x, cond, text, 0.4, False, False)
    mixed = F5HipModel(DiTArch(**TINY), sd, gemm_planes=3).transformer_forward(x, cond, text, 0.4, False, False)
    assert torch.isfinite(mixed).all()
    assert _report("outlier channel, bf16x3", strict, ref) < 1e-3
    assert _report("outlier channel, mixed fp16 (saturating)", mixed, ref) < 1e-3
