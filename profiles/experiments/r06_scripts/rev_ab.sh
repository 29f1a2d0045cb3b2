#!/bin/bash
cd $GRAFT_REPO_ROOT
for round in 1 2 3; do
  for v in HEAD REV; do
    echo -n "$v "; IDOCP_HIP_LIB=$PWD/build/variants/libidocp_hip_$v.so python bench.py --no-cpu-baseline --no-latency --steps 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), {k:round(v,3) for k,v in d['config']['kernel_ms'].items()})"
  done
done
