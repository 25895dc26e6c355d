# rocprofv3 kernel trace of the default bench run and HBM traffic counters of the objective and
# LSTM kernels, under gpurun_out/$ROUND/ (summaries are copied into profiles/ by hand).
set -e
ROUND=${ROUND:-r03}
mkdir -p gpurun_out/$ROUND
export TMPDIR=/tmp
R=$PWD
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$ROUND/prof -o bench --output-format csv -- python3 bench.py --no-cpu-baseline --no-kernel-timing --no-fp32-network --steps 20 --warmup 5 > gpurun_out/$ROUND/bench_prof.json 2> gpurun_out/$ROUND/bench_prof.err
echo trace done
python3 tools/summarize_rocprof.py $(find gpurun_out/$ROUND/prof -name "*kernel_stats.csv" | head -1) 25 > gpurun_out/$ROUND/bench_c2_kernel_stats.md
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/$ROUND/pmc_f_loss -o p --output-format csv -- python3 tools/time_loss_kernels.py 4096 > gpurun_out/$ROUND/pmc_f_loss.log 2>&1
echo pmc1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/$ROUND/pmc_w_loss -o p --output-format csv -- python3 tools/time_loss_kernels.py 4096 > gpurun_out/$ROUND/pmc_w_loss.log 2>&1
echo pmc2
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/$ROUND/pmc_f_lstm -o p --output-format csv -- python3 tools/run_lstm128_once.py 4096 > gpurun_out/$ROUND/pmc_f_lstm.log 2>&1
echo pmc3
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/$ROUND/pmc_w_lstm -o p --output-format csv -- python3 tools/run_lstm128_once.py 4096 > gpurun_out/$ROUND/pmc_w_lstm.log 2>&1
echo pmc4
python3 tools/pmc_traffic.py $(find gpurun_out/$ROUND/pmc_f_lstm gpurun_out/$ROUND/pmc_w_lstm -name "*counter_collection.csv") > gpurun_out/$ROUND/pmc_lstm.json
python3 tools/pmc_traffic.py $(find gpurun_out/$ROUND/pmc_f_loss gpurun_out/$ROUND/pmc_w_loss -name "*counter_collection.csv") > gpurun_out/$ROUND/pmc_loss.json
cat gpurun_out/$ROUND/pmc_lstm.json
