#!/bin/bash
# does the length of the untimed warm-up change bench.py's timed region (clock ramp)?  same box, same build
cd $GRAFT_REPO_ROOT
for w in 40 400 2000 40; do
  timeout -k 10 200 python3 bench.py --steps 400 --warmup $w --no-cpu-baseline --no-end-to-end 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('warmup $w: kernel_us', round(d['roofline']['kernel_us'],2), 'ms_per_step', round(d['ms_per_step']*1e3,2))"
done
for s in 400 2000 9000; do
  timeout -k 10 200 python3 bench.py --steps $s --warmup 400 --no-cpu-baseline --no-end-to-end 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('steps $s warmup 400: kernel_us', round(d['roofline']['kernel_us'],2), 'ms_per_step', round(d['ms_per_step']*1e3,2))"
done
timeout -k 10 120 python3 scripts/variant_check.py --tag vc 2>/dev/null | grep tag
