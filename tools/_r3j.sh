mkdir -p gpurun_out/r3j
python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "gemm or qkv" > gpurun_out/r3j/ops.log 2>&1; echo "ops rc=$?"
python -m pytest tests/test_gpu_configs.py -q -m gpu -x -s -k "c3 or c4_sampler" > gpurun_out/r3j/c3.log 2>&1; echo "c3 rc=$?"
python bench.py --batch 8 --no-cpu-baseline > gpurun_out/r3j/bench_b8.json 2> gpurun_out/r3j/bench_b8.err; echo "bench b8 rc=$?"
python tools/two_stream_probe.py > gpurun_out/r3j/two_stream_probe.txt 2>&1; echo "probe rc=$?"
export TMPDIR=/tmp
rocprofv3 --kernel-trace -d gpurun_out/r3j/prof_b8 -o b8 -- python3 bench.py --batch 8 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r3j/bench_b8_prof.json 2> gpurun_out/r3j/bench_b8_prof.err; echo "prof rc=$?"
python tools/rocpd_stats.py $(find gpurun_out/r3j/prof_b8 -name "*.db" | head -1) > gpurun_out/r3j/rocprof_b8.csv
find gpurun_out/r3j -name "*.db" -delete
