"""Diagnostics: one C2 utterance, 2 Euler steps, printed as a checksum -- run under different F5HIP_WIDE* settings and compare."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tts_indic_server_f5_amd import synth
from tts_indic_server_f5_amd.model import F5TTS_BASE, F5HipModel
m = F5HipModel(F5TTS_BASE, synth.dit_state_dict(), gemm_planes=3)
gc = torch.Generator().manual_seed(14)
cond = torch.randn(1, 469, 100, generator=gc)
out, _ = m.sample(cond, synth.text_ids(), 1404, steps=2, cfg_strength=2.0, sway_sampling_coef=-1.0, y0=synth.noise(1404, 0)[None])
torch.save(out.cpu(), os.environ.get("OUT", "/tmp/out.pt"))
print("WIDE", os.environ.get("F5HIP_WIDE"), "QKV_ONLY", os.environ.get("F5HIP_WIDE_QKV_ONLY"), "GEN_ONLY", os.environ.get("F5HIP_WIDE_GENERIC_ONLY"), "sum", float(out.double().sum()), "abs", float(out.double().abs().sum()))
