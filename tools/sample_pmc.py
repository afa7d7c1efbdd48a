#!/usr/bin/env python3
"""Two Euler steps of the C2 workload for rocprofv3 --pmc collection (diagnostics)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tts_indic_server_f5_amd import synth
from tts_indic_server_f5_amd.model import F5TTS_BASE, F5HipModel
m = F5HipModel(F5TTS_BASE, synth.dit_state_dict())
g = torch.Generator().manual_seed(14)
cond = torch.randn(1, 469, 100, generator=g)
for _ in range(2):
    out, _ = m.sample(cond, synth.text_ids(), 1404, steps=2, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=1)
torch.cuda.synchronize()
print("done", float(out.abs().mean()))
