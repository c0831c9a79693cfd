#!/bin/bash
# Sweep of the batch-sharing kernel's tuning knobs at cfgA; one process per variant.  Usage: exp_variants.sh <out.jsonl>
OUT="${1:-gpurun_out/variants.jsonl}"
: > "$OUT"
run() { # tag, env...
  local tag=$1; shift
  env "$@" timeout -k 10 120 python3 scripts/variant_check.py --tag "$tag" >> "$OUT" 2>> "$OUT.err" || { echo "FAILED $tag" >> "$OUT"; return 1; }
  tail -n 1 "$OUT" | cut -c1-260
}
run base FFTCONV_X=0 || exit 1
run slot FFTCONV_EXP_SLOT=1 || exit 1
run slot_prio2 FFTCONV_EXP_SLOT=1 FFTCONV_EXP_PRIO=2 || exit 1
run slot_prio1 FFTCONV_EXP_SLOT=1 FFTCONV_EXP_PRIO=1 || exit 1
