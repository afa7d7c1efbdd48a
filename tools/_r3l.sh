mkdir -p gpurun_out/r3l
python -m pytest tests -q -m gpu > gpurun_out/r3l/gpu_tests.log 2>&1; echo "tests rc=$?"
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3l/smoke.log 2>&1; echo "smoke rc=$?"
python bench.py > gpurun_out/r3l/bench_c2.json 2> gpurun_out/r3l/bench_c2.err; echo "bench c2 rc=$?"
python bench.py --cpu-full --steps 5 --warmup 2 > gpurun_out/r3l/bench_c2_cpufull.json 2> gpurun_out/r3l/bench_c2_cpufull.err; echo "bench cpu-full rc=$?"
