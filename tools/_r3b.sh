set -o pipefail
mkdir -p gpurun_out/r3b
cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu -x > gpurun_out/r3b/gpu_tests.log 2>&1; echo "tests rc=$?"
python tools/attn_mode_tapdiff.py > gpurun_out/r3b/tapdiff.txt 2>&1; echo "tapdiff rc=$?"
export TMPDIR=/tmp
rocprofv3 --kernel-trace -d gpurun_out/r3b/prof_b8 -o b8 -- python3 bench.py --batch 8 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r3b/bench_b8_prof.json 2> gpurun_out/r3b/bench_b8_prof.err; echo "prof rc=$?"
python tools/rocpd_stats.py gpurun_out/r3b/prof_b8/*/b8_results.db > gpurun_out/r3b/rocprof_b8.csv 2>gpurun_out/r3b/rocpd.err || python tools/rocpd_stats.py $(find gpurun_out/r3b/prof_b8 -name "*.db" | head -1) > gpurun_out/r3b/rocprof_b8.csv
M=22528 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d gpurun_out/r3b/pmc_gemm -o g -- python3 tools/gemm_pmc.py > gpurun_out/r3b/pmc_gemm.log 2>&1; echo "pmc rc=$?"
python tools/attn_pmc.py --summarise gpurun_out/r3b/pmc_gemm gemm5 > gpurun_out/r3b/pmc_gemm5_sq_m22528.txt 2>&1
python bench.py --batch 8 --no-cpu-baseline > gpurun_out/r3b/bench_b8.json 2> gpurun_out/r3b/bench_b8.err; echo "bench b8 rc=$?"
find gpurun_out/r3b -name "*.db" -size +20M -delete
