#!/bin/bash
# Two independent 256-thread workgroups per CU (FFTCONV_PERS=2): does a start offset between the two co-resident
# workgroups (FFTCONV_STAGGER, units of 64 cycles, workgroups >= 256) bring the first round to the later rounds' rate?
cd $GRAFT_REPO_ROOT
echo "== base (NB=4)"; timeout -k 10 120 python3 scripts/variant_check.py --tag base || exit 1
for s in 0 16 48 96 160 256; do
  echo "== PERS=2 STAGGER=$s"
  FFTCONV_PERS=2 FFTCONV_STAGGER=$s timeout -k 10 120 python3 scripts/variant_check.py --tag "pers2_stagger$s" || exit 1
done
for s in 48 160; do
  echo "== phase profile PERS=2 STAGGER=$s"
  FFTCONV_PERS=2 FFTCONV_STAGGER=$s timeout -k 10 120 python3 scripts/phase_profile.py | grep -E "items=|clock|lifetime|round" || exit 1
done
