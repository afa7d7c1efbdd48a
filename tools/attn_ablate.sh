#!/bin/bash
# Timing ablations of attn3's hot loop (diagnostics; the results are WRONG numerically): one library per A3_ABL value (attn3.h: bit 0 no ring
# step, bit 1 exp2 -> v_mul, bit 2 no LDS fragment reads, bit 3 no P V / row-sum MFMAs, bit 4 no KV loop, bit 5 the launch alone), each = the production objects with tu_attn.hip
# recompiled.  Build here (no GPU needed), then on the GPU box:
#   for n in 1 2 4 8 5 7 15; do F5HIP_LIB=$PWD/tts-indic-server-f5_amd/csrc/abl/libf5hip_abl$n.so F5HIP_TORCH_OPS=0 ATTN_AB_SHAPES=0,1,4 python tools/attn_ab.py; done
# Result of round 3: profiles/r03_attn_ablate.txt.
set -e
cd "$(dirname "$0")/.."
C="tts-indic-server-f5_amd/csrc"; O="$C/_obj"; mkdir -p "$C/abl" /tmp/attn_abl
for n in ${@:-1 2 4 8 5 7 15}; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DA3_ABL=$n -c -o /tmp/attn_abl/tu_attn_$n.o "$C/tu_attn.hip" 2>/dev/null &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$C/abl/libf5hip_abl$n.so" $O/f5hip.o $O/tu_gemm_reg.o $O/tu_gemm3.o $O/tu_gemm5_generic.o $O/tu_gemm5_qkv.o $O/tu_gemm6.o $O/tu_conv5.o /tmp/attn_abl/tu_attn_$n.o && echo "built abl$n" ) &
done
wait
