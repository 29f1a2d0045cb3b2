#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/icache; mkdir -p $out
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $out -o ic -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-latency > $out/ic.log 2>&1
cd $R
python3 profiles/summarize_rocpd.py $out/ic_results.db > $out/r06_anymal_trotting_pmc_icache.txt
rm -f $out/ic_results.db
grep -E "ICACHE|IFETCH|WAIT_INST|WAVE_CYCLES" $out/r06_anymal_trotting_pmc_icache.txt | grep -E "condense<D, false, 6, false, false, true, false>|riccati_backward_reg|forward_expand|nominal<D, true, false>" 
