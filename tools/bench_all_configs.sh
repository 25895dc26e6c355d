#!/bin/bash
# Every benchmark configuration once, results under gpurun_out/$ROUND/ (copied into profiles/ by hand).
ROUND=${ROUND:-r03}
set -e
mkdir -p gpurun_out/$ROUND
run() { name=$1; shift; echo "== $name"; timeout -k 10 "$TMO" python bench.py "$@" > gpurun_out/$ROUND/bench_$name.json 2> gpurun_out/$ROUND/bench_$name.err; python tools/show_bench.py gpurun_out/$ROUND/bench_$name.json; }
TMO=400 run c2 --config c2
TMO=200 run c1 --config c1
TMO=300 run c3 --config c3
TMO=200 run c3_mlp --config c3 --net mlp --no-cpu-baseline
TMO=500 run c4 --config c4
TMO=300 run c5 --config c5
TMO=200 run c2_separate --config c2 --separate-kernels --no-cpu-baseline
timeout -k 10 200 python tools/time_loss_kernels.py 2048 4096 8192 16384 65536 > gpurun_out/$ROUND/loss_kernels.txt 2>&1
cat gpurun_out/$ROUND/loss_kernels.txt
