#!/bin/bash
# Sweep of the batch-sharing kernel's tuning knobs at cfgA; one process per variant.  Usage: exp_variants.sh <out.jsonl>
OUT="${1:-gpurun_out/variants.jsonl}"
: > "$OUT"
run() { # tag, env...
  local tag=$1; shift
  env "$@" timeout -k 10 120 python3 scripts/variant_check.py --tag "$tag" >> "$OUT" 2>> "$OUT.err" || { echo "FAILED $tag" >> "$OUT"; return 1; }
  tail -n 1 "$OUT"
}
run base FFTCONV_X=0 || exit 1
for s in 2 4 8 12 16 24; do run sleep$s FFTCONV_EXP_SLEEP=$s || exit 1; done
run prio1 FFTCONV_EXP_PRIO=1 || exit 1
run prio2 FFTCONV_EXP_PRIO=2 || exit 1
for p in 2 3 4 13 22 30 40; do run pref$p FFTCONV_EXP_PREF=$p || exit 1; done
run pref40_sleep8 FFTCONV_EXP_PREF=40 FFTCONV_EXP_SLEEP=8 || exit 1
run pref22_sleep8 FFTCONV_EXP_PREF=22 FFTCONV_EXP_SLEEP=8 || exit 1
run pref40_prio2 FFTCONV_EXP_PREF=40 FFTCONV_EXP_PRIO=2 || exit 1
run pref40_prio1 FFTCONV_EXP_PREF=40 FFTCONV_EXP_PRIO=1 || exit 1
