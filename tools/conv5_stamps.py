#!/usr/bin/env python3
"""Where a conv5 workgroup spends its cycles (diagnostics kernel, csrc/conv5.h): python tools/conv5_stamps.py [PREC] [K] [BATCH]
Stage-1 BigVGAN shape (768 -> 768 channels, 4096 rows per item).  Prints per-k-step cycles of the consumer and the loader wave, and the
share each spends waiting: consumer at barriers; loader in its counted vmcnt waits, at barriers, in window stores, in issue + fetch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tts_indic_server_f5_amd import ops
prec = int(sys.argv[1]) if len(sys.argv) > 1 else 3
k = int(sys.argv[2]) if len(sys.argv) > 2 else 7
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 4
ci = co = int(os.environ.get("C", 768))
P = int(os.environ.get("P", 4096))
dil = int(os.environ.get("DIL", 1))
x = torch.randn(batch * P, ci).cuda()
w = torch.randn(co, ci, k) / (ci * k) ** 0.5
for impl in (0, 5):
    _, us, st = ops.conv1d(x, w, None, None, batch=batch, valid=P - 88 * (P // 1024), dilation=dil, prec=prec, impl=impl, iters=20, stamps=impl == 5)
    fl = 2.0 * batch * P * ci * co * k
    print(f"impl {impl} prec {prec} C {ci} k {k} dil {dil} batch {batch}: {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s useful", flush=True)
st = st.astype(np.float64)
chunk_c = 64 if prec == 3 else 32
nk = (ci // chunk_c) * k
c, l = st[:, 0:5], st[:, 8:15]
ok = c[:, 2] > 0
c, l = c[ok], l[ok]
print(f"blocks {ok.sum()}  k-steps per tile {nk}")
med = lambda a: float(np.median(a))
print(f"consumer: prologue {med(c[:,1]-c[:,0]):8.0f} cyc | loop {med(c[:,2]-c[:,1]):9.0f} cyc = {med(c[:,2]-c[:,1])/nk:7.1f} / k-step | at barriers {med(c[:,3]):9.0f} = {med(c[:,3])/nk:6.1f} / k-step ({100*med(c[:,3]/(c[:,2]-c[:,1])):4.1f} %) | epilogue {med(c[:,4]-c[:,2]):7.0f}")
print(f"loader:   prologue {med(l[:,1]-l[:,0]):8.0f} cyc | loop {med(l[:,2]-l[:,1]):9.0f} cyc = {med(l[:,2]-l[:,1])/nk:7.1f} / k-step | counted waits {med(l[:,3])/nk:6.1f} | at barriers {med(l[:,4])/nk:6.1f} | window stores {med(l[:,5])/nk:6.1f} | issue + fetch {med(l[:,6])/nk:6.1f}   (cycles / k-step)")
