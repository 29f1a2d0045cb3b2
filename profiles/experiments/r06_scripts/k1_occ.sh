#!/bin/bash
# K1 (un_linearize) against the LDS it holds: the committed kernel, + 7 000 B (six wavefronts per CU), + 13 000 B (four)
cd $GRAFT_REPO_ROOT
for r in 1 2; do for v in k1pad0 k1pad7 k1pad13; do
IDOCP_HIP_LIB=$PWD/build/variants/libidocp_hip_$v.so python bench.py --workload iiwa14 --steps 10 --warmup 3 --no-cpu-baseline --no-latency 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', 'step %.3f'%d['ms_per_step'], {k:round(v,3) for k,v in d['config']['kernel_ms'].items()})"
done; done
