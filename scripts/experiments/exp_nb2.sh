#!/bin/bash
# NB = 2 / 256-thread workgroups (two independent workgroups per CU) against the shipped NB = 4 / 512-thread build
cd $GRAFT_REPO_ROOT
for v in "" "FFTCONV_PERS=2" "FFTCONV_PERS=4" "FFTCONV_PERS=2 FFTCONV_TILE=2048"; do
  echo "== $v"
  env $v timeout -k 10 120 python3 scripts/variant_check.py --tag "$v" || exit 1
done
echo "== phase profile PERS=2"
FFTCONV_PERS=2 timeout -k 10 120 python3 scripts/phase_profile.py || exit 1
