#!/usr/bin/env python3
"""A few launches of the attention kernel at the C2 / C3 shapes for rocprofv3 --pmc collection (diagnostics):
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES \
            SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d gpurun_out/pmc_attn -o a -- python3 tools/attn_pmc.py
then `python tools/attn_pmc.py --summarise gpurun_out/pmc_attn [kernel-name substring, default attn3]` prints per-kernel means."""
import csv
import glob
import os
import sys

if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
    files = glob.glob(os.path.join(sys.argv[2], "**", "*counter_collection.csv"), recursive=True)
    acc = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            if (sys.argv[3] if len(sys.argv) > 3 else "attn3") not in k:
                continue
            key = (k.split("(")[0][-60:], r["Grid_Size"], r["Counter_Name"])
            a = acc.setdefault(key, [0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    for (k, g, c), (n, v) in sorted(acc.items()):
        print(f"{k:60s} grid {g:>8s} {c:28s} mean {v / n:16.0f}  ({n} dispatches)")
    sys.exit(0)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tts_indic_server_f5_amd import ops  # noqa: E402

for lens, heads in (((1404, 1404), 16), ((1404,) * 16, 16)):
    n, D = sum(lens), 64 * heads
    g = torch.Generator().manual_seed(1)
    q, k, v = (torch.randn(n, D, generator=g).cuda() for _ in range(3))
    ops.attention(q, k, v, lens, heads=heads, impl=3, iters=10)
torch.cuda.synchronize()
