import os, sys
sys.path.insert(0, "/root/repo")
import torch
from tts_indic_server_f5_amd import synth
from tts_indic_server_f5_amd.vocoder import F5HipBigVGAN
bv = F5HipBigVGAN(synth.bigvgan_state_dict())
mel = (torch.randn(1, 100, 936) * 1.5 - 1.0).cuda()
for _ in range(3):
    w = bv(mel)
torch.cuda.synchronize()
print("done", tuple(w.shape))
