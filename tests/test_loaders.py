"""CPU: the file-level loaders with the reference's names and signatures (F/infer/utils_infer.py:92-130,175-260) round-trip synthetic
checkpoint FILES written with the reference's key names -- `ema_model.transformer.*` + `initted` / `step` + the two legacy mel buffers
for the sampler, `weight_g` / `weight_v` under `{"generator": ...}` for BigVGAN, vocos' config.yaml + pytorch_model.bin -- into the
constructor arguments of the HIP objects (recorded by stand-ins: no GPU here; the GPU twin is tests/test_gpu_e2e.py)."""
import json
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tts_indic_server_f5_amd import infer, loaders, synth  # noqa: E402

TINY = dict(dim=128, depth=2, heads=2, ff_mult=2, text_dim=64, conv_layers=2)


class Recorder:
    def __init__(self, *a, **k):
        self.a, self.k = a, k


@pytest.fixture
def recorders(monkeypatch):
    for name in ("F5HipModel", "F5HipVocos", "F5HipBigVGAN"):
        monkeypatch.setattr(loaders, name, type(name, (Recorder,), {}))
    monkeypatch.setattr(loaders, "isinstance", isinstance, raising=False)


def _vocab(tmp_path, n=40):
    p = tmp_path / "vocab.txt"
    p.write_text("".join(chr(33 + i) + "\n" for i in range(n)), encoding="utf-8")
    return str(p)


def _ema_checkpoint(sd):
    ema = {"ema_model." + k: v for k, v in sd.items()}
    ema["initted"] = torch.tensor(True)
    ema["step"] = torch.tensor(1200000)
    ema["ema_model.mel_spec.mel_stft.mel_scale.fb"] = torch.zeros(513, 100)
    ema["ema_model.mel_spec.mel_stft.spectrogram.window"] = torch.zeros(1024)
    return ema


@pytest.mark.parametrize("kind", ["pt", "safetensors"])
def test_load_model_then_load_checkpoint_like_the_reference(tmp_path, recorders, kind):
    sd = synth.dit_state_dict(text_num_embeds=40, **TINY)
    ema = _ema_checkpoint(sd)
    if kind == "pt":
        path = str(tmp_path / "model_1200000.pt")
        torch.save({"ema_model_state_dict": ema, "model_state_dict": {k: v + 1 for k, v in sd.items()}}, path)
    else:
        from safetensors.torch import save_file
        path = str(tmp_path / "model_1200000.safetensors")
        save_file({k: v.contiguous() for k, v in ema.items()}, path)
    model = infer.load_model(infer.DiT, TINY, mel_spec_type="vocos", vocab_file=_vocab(tmp_path), ode_method="midpoint")
    assert isinstance(model, loaders.UnloadedModel) and model.arch.text_num_embeds == 40 and model.arch.mel_dim == 100
    with pytest.raises(RuntimeError, match="no weights"):
        model.sample(None, None, 10)
    loaded = infer.load_checkpoint(model, path, "cuda", use_ema=True)
    arch, got = loaded.a
    assert arch == model.arch and loaded.k["odeint_kwargs"] == dict(method="midpoint") and loaded.k["mel_spec_type"] == "vocos"
    assert loaded.k["vocab_char_map"]["!"] == 0 and str(loaded.k["device"]) == "cuda:0"
    assert set(got) == set(sd)                                   # prefix stripped; initted / step / the two legacy mel buffers dropped
    assert all(torch.equal(got[k], sd[k]) for k in sd)
    if kind == "pt":                                             # use_ema=False takes model_state_dict as it stands
        raw = infer.load_checkpoint(model, path, "cuda", use_ema=False)
        assert torch.equal(raw.a[1]["transformer.proj_out.bias"], sd["transformer.proj_out.bias"] + 1)
    # one call, like the reference's callers that still pass the checkpoint to load_model
    again = infer.load_model(infer.DiT, TINY, vocab_file=_vocab(tmp_path), ckpt_path=path)
    assert torch.equal(again.a[1]["transformer.proj_out.weight"], sd["transformer.proj_out.weight"])


def test_load_model_backbone_tags_and_errors(tmp_path, recorders):
    v = _vocab(tmp_path)
    assert isinstance(infer.load_model(infer.UNetT, dict(dim=128, depth=4, heads=2, ff_mult=4), vocab_file=v).arch, loaders.UNetTArch)
    assert isinstance(infer.load_model(infer.MMDiT, dict(dim=128, depth=3, heads=2, ff_mult=2), vocab_file=v).arch, loaders.MMDiTArch)
    with pytest.raises(ValueError):
        infer.load_model(infer.DiT, TINY)                        # the reference's packaged vocab.txt is not shipped
    with pytest.raises(TypeError):
        infer.load_checkpoint(object(), "x.pt", "cuda")
    with pytest.raises(RuntimeError, match="no hub download"):
        infer.load_vocoder("vocos", is_local=False)
    with pytest.raises(ValueError):
        infer.load_vocoder("wavenet", is_local=True, local_path=str(tmp_path))


def test_load_vocoder_vocos_local_files(tmp_path, recorders):
    import yaml
    sd = synth.vocos_state_dict()
    sd["feature_extractor.mel_spec.spectrogram.window"] = torch.hann_window(1024)
    torch.save(sd, str(tmp_path / "pytorch_model.bin"))
    cfg = {"feature_extractor": {"class_path": "vocos.feature_extractors.MelSpectrogramFeatures",
                                 "init_args": {"sample_rate": 24000, "n_fft": 1024, "hop_length": 256, "n_mels": 100, "padding": "center"}},
           "backbone": {"class_path": "vocos.models.VocosBackbone", "init_args": {"input_channels": 100, "dim": 512, "intermediate_dim": 1536, "num_layers": 8}},
           "head": {"class_path": "vocos.heads.ISTFTHead", "init_args": {"dim": 512, "n_fft": 1024, "hop_length": 256, "padding": "center"}}}
    (tmp_path / "config.yaml").write_text(yaml.safe_dump(cfg))
    voc = infer.load_vocoder("vocos", is_local=True, local_path=str(tmp_path), device="cuda")
    assert type(voc).__name__ == "F5HipVocos"
    assert voc.k == dict(device=torch.device("cuda:0"), in_channels=100, dim=512, intermediate_dim=1536, num_layers=8, n_fft=1024, hop_length=256)
    assert torch.equal(voc.a[0]["backbone.embed.weight"], sd["backbone.embed.weight"])


def test_load_vocoder_bigvgan_weight_norm_checkpoint(tmp_path, recorders):
    sd = synth.bigvgan_state_dict(upsample_rates=(4, 2), upsample_kernel_sizes=(8, 4), upsample_initial_channel=32, resblock_kernel_sizes=(3,))
    wn = {}
    for k, v in sd.items():                                        # the generator as its checkpoints store it: weight-norm parameters
        if k.endswith(".weight") and v.ndim == 3:
            n = v.flatten(1).norm(dim=1).view(-1, 1, 1)
            wn[k + "_g"], wn[k + "_v"] = n.clone(), v.clone()
        else:
            wn[k] = v
    torch.save({"generator": wn}, str(tmp_path / "bigvgan_generator.pt"))
    h = dict(num_mels=100, upsample_rates=[4, 2], upsample_kernel_sizes=[8, 4], upsample_initial_channel=32, resblock_kernel_sizes=[3],
             resblock_dilation_sizes=[[1, 3, 5]], activation="snakebeta", snake_logscale=True)
    (tmp_path / "config.json").write_text(json.dumps(h))
    voc = infer.load_vocoder("bigvgan", is_local=True, local_path=str(tmp_path), device="cuda:0")
    assert type(voc).__name__ == "F5HipBigVGAN" and voc.k["upsample_rates"] == (4, 2) and voc.k["resblock_dilation_sizes"] == ((1, 3, 5),)
    assert any(k.endswith("weight_g") for k in voc.a[0]) and len(voc.a[0]) == len(wn)     # folded by F5HipBigVGAN itself, like remove_weight_norm()
