#!/bin/bash
# end-of-session measurement set: cfgA trace + PMC passes, the bench line, the same-GPU yardstick sweep, the many-channel
# table with per-kernel times, and kernel stats + HBM traffic counters of the other BASELINE configurations
cd $GRAFT_REPO_ROOT
bash scripts/gpu_prof.sh r2f || exit 1
timeout -k 10 400 python3 bench.py > gpurun_out/r2f_bench.json 2> gpurun_out/r2f_bench.err || exit 1
timeout -k 10 300 python3 scripts/sweep_vs_rocfft.py > gpurun_out/r2f_sweep.jsonl 2> gpurun_out/r2f_sweep.err || exit 1
timeout -k 10 300 python3 scripts/dense_check.py > gpurun_out/r2f_dense.jsonl 2>&1 || exit 1
bash scripts/experiments/prof_dense.sh 0 > gpurun_out/r2f_dense0_prof.txt 2>&1 || exit 1
bash scripts/experiments/prof_dense.sh 4 > gpurun_out/r2f_dense4_prof.txt 2>&1 || exit 1
for cfg in cfg0 cfgB cfgC cfgD; do
  bash scripts/gpu_prof_traffic.sh r2f_$cfg --config $cfg > gpurun_out/r2f_${cfg}_traffic.log 2>&1 || exit 1
done
