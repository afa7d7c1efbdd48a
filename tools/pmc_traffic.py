#!/usr/bin/env python3
"""Memory-side traffic per launch of the block-GEMM kernel from two rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE cannot share a pass:
MI355X_MICROARCH.md, rocprofv3 PMC slots), corrected as that guide prescribes (gfx950: FETCH_SIZE counts 64 B per 128-B request ->
x2; both counters are in KiB... as reported by rocprofv3 in this image: see `unit` below).

  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_f -o f -- python3 tools/sample_pmc.py
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_w -o w -- python3 tools/sample_pmc.py
  python tools/pmc_traffic.py gpurun_out/pmc_f/f_results.db gpurun_out/pmc_w/w_results.db > profiles/r03_pmc_hbm_traffic.json"""
import json
import sqlite3
import sys


def per_kernel(db_path, counter):
    db = sqlite3.connect(db_path)
    cols = [c[1] for c in db.execute("pragma table_info('pmc_events')")]
    # pmc_events view: one row per (dispatch, counter) with the kernel's name
    name_col = "name" if "name" in cols else cols[0]
    rows = db.execute("select * from pmc_events").fetchall()
    idx = {c: i for i, c in enumerate(cols)}
    out = {}
    for r in rows:
        cname = r[idx.get("counter_name", idx.get("pmc_name", 0))]
        if cname != counter:
            continue
        kname = r[idx.get("name", idx.get("kernel_name", 0))]
        val = float(r[idx.get("value", idx.get("counter_value", 0))])
        a = out.setdefault(kname, [0, 0.0])
        a[0] += 1
        a[1] += val
    return out, cols


def main(fetch_db, write_db):
    f, cols = per_kernel(fetch_db, "FETCH_SIZE")
    w, _ = per_kernel(write_db, "WRITE_SIZE")
    res = {"columns_seen": cols, "kernels": {}}
    tot_launch, tot_bytes = 0, 0.0
    for k in sorted(f):
        if "gemm5_kernel" not in k:
            continue
        n, fs = f[k]
        ws = w.get(k, [n, 0.0])[1]
        # FETCH_SIZE / WRITE_SIZE are reported in KiB by rocprofv3's derived counters; FETCH_SIZE x2 on gfx950 (guide, HBM section)
        fetch_b, write_b = fs * 1024.0 * 2.0 / n, ws * 1024.0 / n
        res["kernels"][k[:120]] = {"launches": n, "fetch_bytes_per_launch_x2": fetch_b, "write_bytes_per_launch": write_b}
        tot_launch += n
        tot_bytes += (fetch_b + write_b) * n
    res["bytes_per_launch"] = tot_bytes / max(tot_launch, 1)
    # every kernel, for the text table (stderr)
    sys.stderr.write("# kernel | launches | fetch MB/launch (x2-corrected) | write MB/launch\n")
    for k in sorted(f, key=lambda k: -f[k][1]):
        n, fs = f[k]
        ws = w.get(k, [n, 0.0])[1]
        sys.stderr.write(f"{k[:70]:70s} {n:6d} {fs * 1024 * 2 / n / 1e6:10.2f} {ws * 1024 / max(w.get(k, [n])[0], 1) / 1e6:10.2f}\n")
    res["source"] = "profiles/r03_pmc_hbm_traffic.json: rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, separate passes over tools/sample_pmc.py, mean over the gemm5 launches (fp16 block GEMMs of the C2 workload)"
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
