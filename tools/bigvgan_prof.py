#!/usr/bin/env python3
"""BigVGAN decode under rocprofv3: rocprofv3 --kernel-trace -d gpurun_out/prof_bv -o bv -- python3 tools/bigvgan_prof.py PLANES BATCH"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tts_indic_server_f5_amd import synth
from tts_indic_server_f5_amd.vocoder import F5HipBigVGAN
planes, batch = int(sys.argv[1]), int(sys.argv[2])
bv = F5HipBigVGAN(synth.bigvgan_state_dict(), gemm_planes=planes)
mel = (torch.randn(batch, 100, 936) * 1.5 - 1.0).cuda()
for _ in range(3):
    w = bv(mel)
torch.cuda.synchronize()
print("done", tuple(w.shape))
