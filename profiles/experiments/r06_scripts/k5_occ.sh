#!/bin/bash
cd $GRAFT_REPO_ROOT
for round in 1 2; do
  IDOCP_HIP_LIB=$PWD/build/variants/libidocp_hip_C.so python scratch/k5_skip.py 2>&1 | tail -1
  IDOCP_HIP_LIB=$PWD/build/variants/libidocp_hip_D.so python scratch/k5_skip.py 2>&1 | tail -1
  echo "D + 5000 B pad (4 WG):"; IDOCP_K5_LDS_PAD=5000 IDOCP_HIP_LIB=$PWD/build/variants/libidocp_hip_D.so python scratch/k5_skip.py 2>&1 | tail -1
  echo "C + 12000 B pad (3 WG):"; IDOCP_K5_LDS_PAD=12000 IDOCP_HIP_LIB=$PWD/build/variants/libidocp_hip_D.so python scratch/k5_skip.py 2>&1 | tail -1
done
