import sys, os
sys.path.insert(0, "/root/repo")
import torch
from tts_indic_server_f5_amd import synth
from tts_indic_server_f5_amd.model import F5TTS_BASE, F5HipModel
m = F5HipModel(F5TTS_BASE, synth.dit_state_dict(), gemm_planes=3)
gc = torch.Generator().manual_seed(14)
cond = torch.randn(1, 469, 100, generator=gc)
text = synth.text_ids()
y0 = synth.noise(1404, 0)[None]
kw = dict(steps=2, cfg_strength=2.0, sway_sampling_coef=-1.0)
one, _ = m.sample(cond, text, 1404, y0=y0, **kw)
one2, _ = m.sample(cond, text, 1404, y0=y0, **kw)
four, _ = m.sample(cond.expand(4, -1, -1), text.expand(4, -1), 1404, y0=y0.expand(4, -1, -1), **kw)
four2, _ = m.sample(cond.expand(4, -1, -1), text.expand(4, -1), 1404, y0=y0.expand(4, -1, -1), **kw)
two, _ = m.sample(cond.expand(2, -1, -1), text.expand(2, -1), 1404, y0=y0.expand(2, -1, -1), **kw)
def r(a, b): return (a.float() - b.float()).pow(2).mean().sqrt().item()
print("impl", os.environ.get("F5HIP_GEMM_IMPL"), "one vs one2", r(one, one2), "four vs four2", r(four, four2), "four[0] vs one", r(four[0], one[0]),
      "four[0] vs four[3]", r(four[0], four[3]), "two[0] vs one", r(two[0], one[0]), "two[1] vs one", r(two[1], one[0]))
