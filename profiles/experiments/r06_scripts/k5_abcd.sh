#!/bin/bash
# A/B of K5 variants inside ONE gpurun call: K5 alone (20 launches, scratch/k5_skip.py), two rounds each, then parity of the candidates
cd $GRAFT_REPO_ROOT
for round in 1 2; do
  for v in "$@"; do
    IDOCP_HIP_LIB=$PWD/build/variants/libidocp_hip_$v.so python scratch/k5_skip.py 2>&1 | tail -1
  done
done
for v in "$@"; do
  [ "$v" = "A" ] && continue
  echo "== parity $v"
  IDOCP_HIP_LIB=$PWD/build/variants/libidocp_hip_$v.so python -m pytest tests/test_ocp_gpu.py tests/test_hybrid_gpu.py -x -q -m gpu 2>&1 | tail -2
done
