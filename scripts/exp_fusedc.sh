#!/bin/bash
# N-d column pass: batch items per workgroup (4 / 2 / 1 -> 1 / 2 / 4 workgroups per CU once the kernel fits 128 VGPRs)
for cfg in cfgB cfgC; do
  for nb in 4 2 1; do
    echo "== $cfg FFTCONV_FUSEDC_NB=$nb"
    FFTCONV_FUSEDC_NB=$nb timeout -k 10 200 python3 bench.py --config $cfg --steps 60 --warmup 10 --no-cpu-baseline --no-end-to-end 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['kernel_us'],1), 'us per step, frac', round(d['roofline']['frac'],4))"
  done
done
