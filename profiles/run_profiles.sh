#!/bin/bash
# Collect the round's rocprofv3 evidence on a GPU box (run through gpurun from the repo root):
#   profiles/run_profiles.sh <round> <workload>         e.g.  profiles/run_profiles.sh 01 anymal
# Pass 1: --kernel-trace --stats of the default bench command.
# Pass 2/3: FETCH_SIZE and WRITE_SIZE in separate --pmc passes (no other trace domains).
# The raw rocpd databases are summarised on the box and removed; the text / json summaries come back under gpurun_out/ and are
# copied to profiles/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
rnd=$1; wl=$2
extra=${BENCH_EXTRA:-}          # further bench.py arguments, e.g. BENCH_EXTRA="--horizon 120"
out=$R/gpurun_out/prof_r$rnd
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out -o ${wl}_trace -- python3 $R/bench.py --workload $wl $extra --steps 5 --warmup 2 --no-cpu-baseline --no-latency > $out/${wl}_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out -o ${wl}_fetch -- python3 $R/bench.py --workload $wl $extra --steps 3 --warmup 1 --no-cpu-baseline --no-latency > $out/${wl}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out -o ${wl}_write -- python3 $R/bench.py --workload $wl $extra --steps 3 --warmup 1 --no-cpu-baseline --no-latency > $out/${wl}_write.log 2>&1
cd $R
line=$(grep -o '{"metric".*' $out/${wl}_trace.log | tail -1)
batch=$(echo "$line" | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['config']['batch_per_gpu'])")
hor=$(echo "$line" | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['config']['horizon'])")
{ echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --workload $wl $extra --steps 5 --warmup 2 --no-cpu-baseline"; echo "# bench line: $line"; python3 profiles/summarize_rocpd.py $out/${wl}_trace_results.db; } > $out/r${rnd}_${wl}_kernel_trace.txt
python3 profiles/summarize_rocpd.py $out/${wl}_fetch_results.db $out/${wl}_write_results.db > $out/r${rnd}_${wl}_pmc_hbm.txt
# Pass 4/5 (optional, `sq` as third argument): SQ occupancy / issue counters behind the latency-bound statements of DESIGN.md
if [ "${3:-}" = "sq" ]; then
  cd /tmp
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $out -o ${wl}_sq1 -- python3 $R/bench.py --workload $wl $extra --steps 3 --warmup 1 --no-cpu-baseline --no-latency > $out/${wl}_sq1.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM -d $out -o ${wl}_sq2 -- python3 $R/bench.py --workload $wl $extra --steps 3 --warmup 1 --no-cpu-baseline --no-latency > $out/${wl}_sq2.log 2>&1
  # matrix-core utilisation of the kernels that run their products on FP64 MFMA (K5, S3): busy cycles of the MFMA pipe next to the CU-busy cycles
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA -d $out -o ${wl}_sq3 -- python3 $R/bench.py --workload $wl $extra --steps 3 --warmup 1 --no-cpu-baseline --no-latency > $out/${wl}_sq3.log 2>&1
  cd $R
  python3 profiles/summarize_rocpd.py $out/${wl}_sq1_results.db $out/${wl}_sq2_results.db $out/${wl}_sq3_results.db > $out/r${rnd}_${wl}_pmc_sq.txt
  python3 profiles/make_sq_json.py $out/${wl}_sq1_results.db $out/${wl}_sq2_results.db $out/${wl}_sq3_results.db $wl $batch $hor $rnd > $out/r${rnd}_pmc_sq_${wl}.json
  rm -f $out/${wl}_sq1_results.db $out/${wl}_sq2_results.db $out/${wl}_sq3_results.db
fi
python3 profiles/make_traffic_json.py $out/${wl}_fetch_results.db $out/${wl}_write_results.db $wl $batch $hor $rnd > $out/r${rnd}_pmc_traffic_${wl}.json
rm -f $out/${wl}_trace_results.db $out/${wl}_fetch_results.db $out/${wl}_write_results.db      # the summaries above are what is kept (gpurun_out/ is capped at 64 MiB)
echo "$line"
