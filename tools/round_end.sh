#!/bin/bash
# Everything a round's measurements are taken from, in one GPU call: the GPU test suite, every
# bench configuration, the kernel trace and traffic counters of the default run, the SQ
# counters of the row-form objective, the randomised parity sweep.  Results under gpurun_out/.
ROUND=${ROUND:-r03}
export ROUND
mkdir -p gpurun_out/$ROUND
timeout -k 10 900 python -u -m pytest tests -x -q -m gpu --timeout 120 --timeout-method thread > gpurun_out/$ROUND/gpu_tests.log 2>&1 || { tail -30 gpurun_out/$ROUND/gpu_tests.log; exit 1; }
tail -2 gpurun_out/$ROUND/gpu_tests.log
bash tools/bench_all_configs.sh > gpurun_out/$ROUND/bench_all.log 2>&1 || { tail -20 gpurun_out/$ROUND/bench_all.log; exit 1; }
grep "ms/step" gpurun_out/$ROUND/bench_all.log
bash tools/profile_round.sh > gpurun_out/$ROUND/profile.log 2>&1 || { tail -20 gpurun_out/$ROUND/profile.log; exit 1; }
bash tools/pmc_row.sh 4096 > gpurun_out/$ROUND/pmc_row.log 2>&1 || { tail -20 gpurun_out/$ROUND/pmc_row.log; exit 1; }
python3 tools/pmc_row_summary.py gpurun_out/pmc_row > gpurun_out/$ROUND/row_sq_counters.txt
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -u tools/stress_parity.py ${STRESS_CASES:-2000} 3 > gpurun_out/$ROUND/stress_parity.txt 2>&1 || { tail -20 gpurun_out/$ROUND/stress_parity.txt; exit 1; }
tail -3 gpurun_out/$ROUND/stress_parity.txt
