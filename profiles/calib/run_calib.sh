#!/bin/bash
# gpurun -- 'bash profiles/calib/run_calib.sh 02'   -> gpurun_out/prof_rNN/rNN_pmc_calibration.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; rnd=$1; out=$R/gpurun_out/prof_r$rnd; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out -o calib_fetch -- $R/profiles/calib/calib_counters > $out/calib_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out -o calib_write -- $R/profiles/calib/calib_counters > $out/calib_write.log 2>&1
cd $R
{ echo "# profiles/calib/calib_counters.hip: every kernel moves 2147483648 bytes (2 GiB) per direction it touches; counters in kB"; python3 profiles/summarize_rocpd.py $out/calib_fetch_results.db $out/calib_write_results.db; } > $out/r${rnd}_pmc_calibration.txt
rm -f $out/calib_fetch_results.db $out/calib_write_results.db
grep -A12 "counter" $out/r${rnd}_pmc_calibration.txt
