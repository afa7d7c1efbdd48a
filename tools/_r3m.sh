mkdir -p gpurun_out/r3m
python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "gemm or qkv" > gpurun_out/r3m/ops.log 2>&1; echo "ops rc=$?"
python -m pytest tests/test_gpu_configs.py -q -m gpu -x -s -k "c3 or c4_sampler or c5_e2base_sample" > gpurun_out/r3m/c3.log 2>&1; echo "c3 rc=$?"
python bench.py --batch 8 --no-cpu-baseline > gpurun_out/r3m/bench_b8.json 2> gpurun_out/r3m/err.txt; echo "rc=$?"
F5HIP_GEMM6_PERSIST=0 python bench.py --batch 8 --no-cpu-baseline > gpurun_out/r3m/bench_b8_nopersist.json 2>> gpurun_out/r3m/err.txt; echo "rc=$?"
python bench.py --batch 8 --no-cpu-baseline > gpurun_out/r3m/bench_b8_2.json 2>> gpurun_out/r3m/err.txt; echo "rc=$?"
WARM=300 python tools/gemm6_stamps.py > gpurun_out/r3m/gemm6_stamps.txt 2>&1; echo "stamps rc=$?"
