#!/bin/bash
cd $GRAFT_REPO_ROOT
for round in 1 2 3; do
  for v in HEAD NOFV; do
    IDOCP_HIP_LIB=$PWD/build/variants/libidocp_hip_$v.so python scratch/k5_skip.py 2>&1 | tail -1
  done
done
