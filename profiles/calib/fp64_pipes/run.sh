#!/bin/bash
# FP64 pipe rates of gfx950 as DESIGN.md 4.0a quotes them.  Run on a GPU box from the repo root:  profiles/calib/fp64_pipes/run.sh > gpurun_out/r04_fp64_pipe_rates.txt
d=$(dirname "$0"); t=${TMPDIR:-/tmp}/fp64_pipes; mkdir -p $t
for f in vector_fma_rate mfma_16x16x4_chain mfma_4x4x4_chain mfma_vector_overlap mfma_4x4x4_lane_map; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $t/$f $d/$f.hip 2>/dev/null && { echo "== $f"; $t/$f; }
done
