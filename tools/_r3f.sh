mkdir -p gpurun_out/r3f
python -m pytest tests -q -m gpu -x > gpurun_out/r3f/gpu_tests.log 2>&1; echo "tests rc=$?"
python bench.py > gpurun_out/r3f/bench_c2.json 2> gpurun_out/r3f/bench_c2.err; echo "bench c2 rc=$?"
python bench.py --arch e2 --nfe 64 --batch 8 --steps 3 --warmup 1 > gpurun_out/r3f/bench_c5.json 2> gpurun_out/r3f/bench_c5.err; echo "bench c5 rc=$?"
python bench.py --batch 16 --vocoder bigvgan --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3f/bench_c4.json 2> gpurun_out/r3f/bench_c4.err; echo "bench c4 rc=$?"
