import ctypes as C, sys, time
sys.path.insert(0, "/root/repo")
import torch
from tts_indic_server_f5_amd import _lib, synth
from tts_indic_server_f5_amd.model import F5TTS_BASE, F5HipModel
def counter(n):
    v = C.c_int64(0); _lib.check(_lib.lib().f5hip_get_counter(n.encode(), C.byref(v)), "c"); return v.value
m = F5HipModel(F5TTS_BASE, synth.dit_state_dict())
cond = torch.randn(1, 468, 100, generator=torch.Generator().manual_seed(5)); text = synth.text_ids(60, 240); y0 = [synth.noise(1404, 0)]
kw = dict(steps=4, cfg_strength=2.0, sway_sampling_coef=-1.0, y0=y0)
out, _ = m.sample(cond, text, 1404, **kw)
torch.cuda.synchronize()
print("fused launches", counter("ln_fused"), "timeouts", counter("ln_fuse_timeouts"), "out rms", float(out.float().pow(2).mean().sqrt()), flush=True)
torch.save(out.cpu(), sys.argv[1])
kw["steps"] = 32
for _ in range(2): m.sample(cond, text, 1404, **kw)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): m.sample(cond, text, 1404, **kw)
torch.cuda.synchronize(); print(f"32-step sample: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms", flush=True)
