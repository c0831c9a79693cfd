#!/bin/bash
# cfgB / cfgC / cfgD per-kernel times (rocprofv3 --kernel-trace --stats) of the current build
cd /tmp && export TMPDIR=/tmp
for cfg in cfgB cfgC cfgD; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2g_${cfg}_trace -- python3 $GRAFT_REPO_ROOT/bench.py --config $cfg --steps 30 --warmup 5 --no-cpu-baseline --no-end-to-end > $GRAFT_REPO_ROOT/gpurun_out/r2g_${cfg}.log 2>&1 || exit 1
  echo "== $cfg"
  python3 $GRAFT_REPO_ROOT/scripts/summarize_prof.py $GRAFT_REPO_ROOT/gpurun_out r2g_${cfg} | grep "fc::" | cut -c1-110
  grep '"metric"' $GRAFT_REPO_ROOT/gpurun_out/r2g_${cfg}.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', round(d['roofline']['kernel_us'],1), 'us  frac', round(d['roofline']['frac'],4))"
done
