mkdir -p gpurun_out/r3d
python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "gemm or qkv" > gpurun_out/r3d/ops.log 2>&1; echo "ops rc=$?"
python -m pytest tests/test_gpu_configs.py -q -m gpu -x -s -k "c3 or c4_sampler" > gpurun_out/r3d/c3.log 2>&1; echo "c3 rc=$?"
python tools/gemm6_stamps.py > gpurun_out/r3d/gemm6_stamps.txt 2>&1; echo "stamps rc=$?"
python bench.py --batch 8 --no-cpu-baseline > gpurun_out/r3d/bench_b8.json 2> gpurun_out/r3d/bench_b8.err; echo "bench b8 rc=$?"
