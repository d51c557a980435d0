#!/bin/bash
# evidence for DESIGN section 6: exchange form 0 (flagged words in front of the next launch) against form 1 (separate launches)
cd "$GRAFT_REPO_ROOT"
bash tools/xch_ab.sh > gpurun_out/xch_ab.log 2>&1 || { tail -5 gpurun_out/xch_ab.log; exit 1; }
grep -E "passed|form env" gpurun_out/xch_ab.log
bash tools/xch_phases.sh > gpurun_out/xch_phases.log 2>&1 || { tail -5 gpurun_out/xch_phases.log; exit 1; }
cat gpurun_out/xch_phases.log
timeout -k 10 600 python tools/xch_multirank.py 2>&1 | grep -E "^ranks|rror|Traceback" > gpurun_out/xch_multirank.log
cat gpurun_out/xch_multirank.log
bash tools/xch_prof.sh 0 > /dev/null 2>&1 && cp gpurun_out/xch_prof_0/x_kernel_stats.csv gpurun_out/xch_form0_kernel_stats.csv
bash tools/xch_prof.sh 1 > /dev/null 2>&1 && cp gpurun_out/xch_prof_1/x_kernel_stats.csv gpurun_out/xch_form1_kernel_stats.csv
ls gpurun_out/*.csv
