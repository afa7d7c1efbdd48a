mkdir -p gpurun_out/r3h
export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/r3h/pmc_f -o f -- python3 tools/sample_pmc.py > gpurun_out/r3h/pmc_f.log 2>&1; echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/r3h/pmc_w -o w -- python3 tools/sample_pmc.py > gpurun_out/r3h/pmc_w.log 2>&1; echo "write rc=$?"
python tools/pmc_traffic.py $(find gpurun_out/r3h/pmc_f -name "*.db" | head -1) $(find gpurun_out/r3h/pmc_w -name "*.db" | head -1) > gpurun_out/r3h/pmc_hbm_traffic.json 2> gpurun_out/r3h/pmc_hbm_traffic.txt; echo "traffic rc=$?"
find gpurun_out/r3h -name "*.db" -delete
