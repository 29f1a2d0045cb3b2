#!/bin/bash
cd $GRAFT_REPO_ROOT
for round in 1 2 3; do
  for v in 0 1; do
    echo "IDOCP_SIDE_STREAM_K5=$v"; IDOCP_SIDE_STREAM_K5=$v python bench.py --no-cpu-baseline --no-latency --steps 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), d['config']['step_ms_hip_events']['median'], {k:round(v,3) for k,v in d['config']['kernel_ms'].items()})"
  done
done
for wl in anymal_running; do for v in 0 1; do echo "$wl K5 split $v"; IDOCP_SIDE_STREAM_K5=$v python bench.py --workload $wl --no-cpu-baseline --no-latency --steps 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), {k:round(v,3) for k,v in d['config']['kernel_ms'].items()})"; done; done
python -m pytest tests/test_hybrid_gpu.py tests/test_ocp_gpu.py tests/test_forward_expand_gpu.py tests/test_golden_kkt.py -x -q -m gpu 2>&1 | tail -2
