#!/bin/bash
# Round-3 measurement set, in parts that each fit one gpurun call.  Usage: r3_measure.sh <part> ; outputs under gpurun_out/r3f_*
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT=gpurun_out
mkdir -p $OUT
case "$1" in
  a)  # headline: rocprofv3 trace + every PMC pass, then the bench lines (default steps, and the driver's 20 / 5)
    bash scripts/gpu_prof.sh r3f || exit 1
    timeout -k 10 400 python3 bench.py > $OUT/r3f_bench_cfgA.json 2> $OUT/r3f_bench_cfgA.err || exit 1
    timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $OUT/r3f_bench_cfgA_driver.json 2>> $OUT/r3f_bench_cfgA.err || exit 1
    ;;
  b)  # other BASELINE configurations and the per-GPU shards: kernel stats + FETCH / WRITE passes, then the plain bench line
    for cfg in cfg0 cfgB cfgC cfgD cfgA_shard cfgB_shard cfgC_shard; do
      bash scripts/gpu_prof_traffic.sh r3f_$cfg --config $cfg > $OUT/r3f_${cfg}_traffic.log 2>&1 || exit 1
      timeout -k 10 300 python3 bench.py --config $cfg --steps 100 --warmup 20 --no-cpu-baseline --no-end-to-end > $OUT/r3f_bench_$cfg.json 2> $OUT/r3f_bench_$cfg.err || exit 1
    done
    ;;
  c)  # rows next to the headline path, README shapes, same-GPU yardsticks, float64
    timeout -k 10 300 python3 scripts/bench_next_rows.py > $OUT/r3f_next_rows.jsonl 2> $OUT/r3f_next_rows.err || exit 1
    timeout -k 10 400 python3 scripts/bench_readme_shapes.py > $OUT/r3f_readme_shapes.jsonl 2> $OUT/r3f_readme.err || exit 1
    timeout -k 10 300 python3 scripts/sweep_vs_rocfft.py > $OUT/r3f_sweep.jsonl 2> $OUT/r3f_sweep.err || exit 1
    timeout -k 10 200 python3 scripts/f64_check.py > $OUT/r3f_f64.txt 2>&1 || exit 1
    for cfg in readme1d readme2d readme3d; do
      timeout -k 10 200 python3 bench.py --config $cfg --steps 100 --warmup 20 --no-cpu-baseline > $OUT/r3f_bench_$cfg.json 2> $OUT/r3f_bench_$cfg.err || exit 1
    done
    ;;
  d)  # the whole GPU suite + the driver's smoke
    timeout -k 10 1000 python3 -m pytest tests -q -m gpu -x > $OUT/r3f_gputests.log 2>&1; echo "pytest rc=$?" >> $OUT/r3f_gputests.log
    timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/r3f_smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/r3f_smoke.log
    tail -3 $OUT/r3f_gputests.log; tail -2 $OUT/r3f_smoke.log
    ;;
esac
