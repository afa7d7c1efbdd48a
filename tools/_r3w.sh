mkdir -p gpurun_out/r3w
run() { echo "## $1"; shift; env "$@" python tools/attn_ab.py 2>&1 | grep -v amdgpu.ids; }
{
run new X=1
run prev F5HIP_LIB=$PWD/tts-indic-server-f5_amd/csrc/libf5hip_exp.so F5HIP_TORCH_OPS=0
run new_again X=1
} > gpurun_out/r3w/attn_wait_ab.txt 2>&1
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "attn or attention" > gpurun_out/r3w/test_attn.txt 2>&1; tail -3 gpurun_out/r3w/test_attn.txt
