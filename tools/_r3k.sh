mkdir -p gpurun_out/r3k
python -m pytest tests/test_gpu_ops.py tests/test_gpu_dit.py -q -m gpu -x -k "layernorm or base_sample or tiny" > gpurun_out/r3k/t.log 2>&1; echo "tests rc=$?"
for i in 1 2; do
python bench.py --no-cpu-baseline > gpurun_out/r3k/bench_c2_xcd_$i.json 2> gpurun_out/r3k/err.txt; echo "rc=$?"
F5HIP_LN_XCD=0 python bench.py --no-cpu-baseline > gpurun_out/r3k/bench_c2_noxcd_$i.json 2>> gpurun_out/r3k/err.txt; echo "rc=$?"
done
python bench.py --batch 8 --no-cpu-baseline > gpurun_out/r3k/bench_b8_xcd.json 2>> gpurun_out/r3k/err.txt; echo "rc=$?"
F5HIP_LN_XCD=0 python bench.py --batch 8 --no-cpu-baseline > gpurun_out/r3k/bench_b8_noxcd.json 2>> gpurun_out/r3k/err.txt; echo "rc=$?"
