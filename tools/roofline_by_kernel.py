#!/usr/bin/env python3
"""Per-kernel roofline fractions from a rocprofv3 kernel-trace summary (the CSV tools/rocpd_stats.py writes), so that the fractions quoted in
DESIGN.md / VERDICT.md can be reproduced without a calculator.

  python tools/roofline_by_kernel.py profiles/r03_rocprof_kernel_stats.csv --batch 1 > profiles/r03_roofline_by_kernel.txt
  python tools/roofline_by_kernel.py profiles/r03_rocprof_kernel_stats_c3_share_b8.csv --batch 8 >> profiles/r03_roofline_by_kernel.txt

Algorithmic work per launch (real rows N = 1404 per sequence, two CFG branches per utterance, F5-TTS-Base):
  block GEMMs      2 M N K FLOP with M = 2 B 1404; which GEMM a launch is follows from its tile template arguments and grid
                   (N = 3072 QKV | 2048 FF1 | 1024 out and FF2: these two share one kernel, so the row shows their mean K = 1536)
  attention        4 n^2 64 FLOP per (sequence, head)
  LayerNorm        rows x 1024 x (4 B read + 2 B fp16 plane written): HBM / Infinity-Cache bound, priced against 8 TB/s
Peaks: 2.5 PFLOP/s dense fp16 / bf16 MFMA, 8 TB/s HBM (MI355X_MICROARCH.md).  Durations are rocprof's per-dispatch averages."""
import argparse
import csv
import re

PEAK_TF, PEAK_GBS = 2500.0, 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--batch", type=int, default=1, help="utterances per sampler call (1 = C2, 8 = C3 per-GPU share, 16 = C4)")
    ap.add_argument("--frames", type=int, default=1404)
    ap.add_argument("--dim", type=int, default=1024)
    ap.add_argument("--heads", type=int, default=16)
    ap.add_argument("--cus", type=int, default=256)
    a = ap.parse_args()
    n_seq = 2 * a.batch
    m_real, m_pad = n_seq * a.frames, n_seq * ((a.frames + 127) // 128 * 128)
    rows = list(csv.DictReader(open(a.csv)))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    qkv_calls = max([int(r["Calls"]) for r in rows if re.match(r"gemm6_kernel<true, [1-9]", r["Name"])] or [0])
    print(f"# {a.csv}: batch {a.batch} ({n_seq} sequences of {a.frames} frames, M = {m_real} real / {m_pad} padded rows); device time in the trace {total / 1e6:.1f} ms")
    print(f"# {'kernel':58s} {'grid':>10s} {'calls':>6s} {'avg us':>9s} {'% time':>7s}  {'work / launch':>16s}  {'achieved':>14s}  frac of roof")
    for r in rows:
        name, grid, calls, avg = r["Name"], r["Grid(workgroups)"], int(r["Calls"]), float(r["AverageNs"]) / 1e3
        pct = float(r["Percentage"])
        g = [int(x) for x in grid.split("x")]
        wgs = g[0] * g[1] * g[2]
        work = unit = frac = None
        m5 = re.match(r"gemm5_kernel<true, (\d), (\d+), (\d+),", name)
        m6 = re.match(r"gemm6_kernel<true, (\d), (\d)>", name)
        if m5 or m6:
            if m5:
                epi, rb, cb = int(m5.group(1)), int(m5.group(2)), int(m5.group(3))
                tiles_m, bn = -(-m_pad // (16 * rb)), 16 * cb
            else:
                epi, rbw = int(m6.group(1)), int(m6.group(2))
                tiles_m, bn = -(-m_pad // (256 if rbw == 8 else 176)), 256
            if m6 and wgs == a.cus:
                # persistent grid (one workgroup per CU walks the tiles): the grid no longer tells N.  EPI != 0 is the QKV launch; the EPI 0
                # instantiations are told apart by their call count relative to QKV's (1x: FF1, 2x: out + FF2, 3x: all three share the kernel)
                ratio = calls / max(1, qkv_calls)
                tag, nk = ("QKV", 3.0) if epi else (("FF1", 2.0) if ratio < 1.5 else (("out + FF2 (mean)", 1.5) if ratio < 2.5 else ("FF1 + out + FF2 (mean)", 5.0 / 3.0)))
                fl = 2.0 * m_real * a.dim * a.dim * nk
            elif wgs % tiles_m:
                continue
            else:
                n = wgs // tiles_m * bn
                k = 1536 if (n == a.dim and not epi) else a.dim
                tag = "QKV" if epi else ("FF1" if n == 2 * a.dim else ("out + FF2 (mean)" if n == a.dim else f"N = {n}"))
                fl = 2.0 * m_real * n * k
            work, unit, frac = f"{fl / 1e9:9.2f} GFLOP", f"{fl / avg / 1e6:8.1f} TF/s", fl / avg / 1e6 / PEAK_TF
            name = f"{name[:44]} [{tag}]"
        elif name.startswith("attn3_fwd_kernel"):
            fl = 4.0 * a.frames * a.frames * 64 * a.heads * n_seq
            work, unit, frac = f"{fl / 1e9:9.2f} GFLOP", f"{fl / avg / 1e6:8.1f} TF/s", fl / avg / 1e6 / PEAK_TF
        elif name.startswith("ln_kernel<4"):
            by = m_real * a.dim * 6.0
            work, unit, frac = f"{by / 1e6:9.2f} MB   ", f"{by / avg / 1e3:8.1f} GB/s", by / avg / 1e3 / PEAK_GBS
        if work is None:
            if pct >= 0.5:
                print(f"  {name[:58]:58s} {grid:>10s} {calls:6d} {avg:9.2f} {pct:7.2f}")
            continue
        print(f"  {name[:58]:58s} {grid:>10s} {calls:6d} {avg:9.2f} {pct:7.2f}  {work:>16s}  {unit:>14s}  {frac:.3f}")


if __name__ == "__main__":
    main()
