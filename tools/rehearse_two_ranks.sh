#!/bin/bash
# Rehearse bench.py's multi-rank code path on a one-GPU box: two ranks on cuda:0 over gloo.
# A rank that is still running after the limit gets SIGABRT; python's fault handler then
# prints where every thread was.  usage: tools/rehearse_two_ranks.sh OUTDIR [bench.py args]
out=$1; shift
mkdir -p "$out"
export KVAE_BENCH_REHEARSAL=1 WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533
for r in 0 1; do
  RANK=$r LOCAL_RANK=$r timeout -s ABRT -k 10 100 python -X faulthandler bench.py --gpus 2 "$@" \
      > "$out/rank$r.json" 2> "$out/rank$r.err" &
done
wait
tail -n 40 "$out"/rank0.err "$out"/rank1.err
cat "$out/rank0.json"
