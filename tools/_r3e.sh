mkdir -p gpurun_out/r3e
python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "gemm or qkv" > gpurun_out/r3e/ops.log 2>&1; echo "ops rc=$?"
F5HIP_GEMM6=176 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "gemm or qkv" > gpurun_out/r3e/ops176.log 2>&1; echo "ops176 rc=$?"
F5HIP_GEMM6=256 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "gemm or qkv" > gpurun_out/r3e/ops256.log 2>&1; echo "ops256 rc=$?"
python -m pytest tests/test_gpu_configs.py -q -m gpu -x -s -k "c3 or c4_sampler" > gpurun_out/r3e/c3.log 2>&1; echo "c3 rc=$?"
F5HIP_GEMM6=176 python tools/gemm6_stamps.py > gpurun_out/r3e/gemm6_stamps_176.txt 2>&1; echo "stamps rc=$?"
python bench.py --batch 8 --no-cpu-baseline > gpurun_out/r3e/bench_b8.json 2> gpurun_out/r3e/bench_b8.err; echo "bench b8 rc=$?"
F5HIP_GEMM6=256 python bench.py --batch 8 --no-cpu-baseline > gpurun_out/r3e/bench_b8_256.json 2> gpurun_out/r3e/bench_b8_256.err; echo "bench b8 rc=$?"
