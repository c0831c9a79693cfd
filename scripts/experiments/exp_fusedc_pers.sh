#!/bin/bash
# N-d column pass: persistent two-sequences-per-thread variant against the plain 1024-thread kernel
cd $GRAFT_REPO_ROOT
for cfg in cfgB cfgC; do
  for pers in 0 1; do
    echo "== $cfg FFTCONV_FUSEDC_PERS=$pers"
    FFTCONV_FUSEDC_PERS=$pers timeout -k 10 200 python3 bench.py --config $cfg --steps 60 --warmup 10 --no-cpu-baseline --no-end-to-end 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(round(d['roofline']['kernel_us'],1), 'us per step, frac', round(d['roofline']['frac'],4))" || exit 1
  done
done
