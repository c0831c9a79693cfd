#!/bin/bash
# bench lines of all five BASELINE configurations (kernel time + roofline fraction)
cd $GRAFT_REPO_ROOT
for cfg in cfg0 cfgA cfgB cfgC cfgD; do
  timeout -k 10 300 python3 bench.py --config $cfg --steps 60 --warmup 10 --no-cpu-baseline --no-end-to-end 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('$cfg', round(d['roofline']['kernel_us'],1), 'us  frac', round(d['roofline']['frac'],4), ' value', round(d['value'],1))" || exit 1
done
