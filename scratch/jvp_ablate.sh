#!/bin/bash
# gpurun helper: JVP stage time per mid-kernel variant (see exp_jvp_ablate.py)
mkdir -p gpurun_out; out=gpurun_out/${1:-abl}_jvp_variants.log; : > $out
for v in 0 c f; do
  echo "GEO_JVP_MID=$v" >> $out
  GEO_JVP_MID=$v timeout -k 10 120 python scratch/exp_jvp_ablate.py prod >> $out 2>&1 || { echo "variant $v failed rc=$?" >> $out; break; }
done
grep -v amdgpu.ids $out
