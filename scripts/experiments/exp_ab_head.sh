#!/bin/bash
# A/B on one box: the working tree's library against the library built from a given commit's csrc (default HEAD)
REF=${1:-HEAD}
cd $GRAFT_REPO_ROOT
mkdir -p /tmp/ab && rm -rf /tmp/ab/* && cp -r fft_conv_pytorch_amd/csrc /tmp/ab/csrc_new && cp -r include /tmp/ab/include
echo "(this script expects scripts/experiments/ab_old_csrc.tar made from the reference commit)"
mkdir -p /tmp/ab/old/fft_conv_pytorch_amd && tar -xf scripts/experiments/ab_old_csrc.tar -C /tmp/ab/old && cp -r include /tmp/ab/old/include
(cd /tmp/ab/old/fft_conv_pytorch_amd/csrc && make -j16 OUT=/tmp/ab/lib_old.so BUILD=/tmp/ab/build_old > /dev/null 2>&1) || exit 1
for rep in 1 2; do
  echo "== new"; timeout -k 10 120 python3 scripts/variant_check.py --tag new 2>/dev/null | grep tag
  echo "== old"; FFTCONV_LIB=/tmp/ab/lib_old.so timeout -k 10 120 python3 scripts/variant_check.py --tag old 2>/dev/null | grep tag
done
