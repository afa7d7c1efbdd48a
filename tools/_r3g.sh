mkdir -p gpurun_out/r3g
python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "gemm or qkv" > gpurun_out/r3g/ops.log 2>&1; echo "ops rc=$?"
python -m pytest tests/test_gpu_dit.py -q -m gpu -x -k "base_sample or base_forward or small_forward or batch_of_copies or mmdit_mid" > gpurun_out/r3g/dit.log 2>&1; echo "dit rc=$?"
python bench.py --no-cpu-baseline > gpurun_out/r3g/bench_c2_wd.json 2> gpurun_out/r3g/bench_c2_wd.err; echo "bench wd rc=$?"
F5HIP_GEMM5_WD=0 python bench.py --no-cpu-baseline > gpurun_out/r3g/bench_c2_nowd.json 2> gpurun_out/r3g/bench_c2_nowd.err; echo "bench nowd rc=$?"
python bench.py --no-cpu-baseline > gpurun_out/r3g/bench_c2_wd2.json 2> gpurun_out/r3g/bench_c2_wd2.err; echo "bench wd rc=$?"
export TMPDIR=/tmp
rocprofv3 --kernel-trace -d gpurun_out/r3g/prof_c2 -o c2 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r3g/bench_c2_prof.json 2> gpurun_out/r3g/bench_c2_prof.err; echo "prof rc=$?"
python tools/rocpd_stats.py $(find gpurun_out/r3g/prof_c2 -name "*.db" | head -1) > gpurun_out/r3g/rocprof_c2.csv
find gpurun_out/r3g -name "*.db" -delete
