#!/bin/bash
# cfgA with pieces of the batch-sharing kernel switched off (FC_DIAG builds: libfftconv_diagN.so next to the product library;
# build: make BUILD=/tmp/build_diagN OUT=.../libfftconv_diagN.so EXTRA="-fno-slp-vectorize -DFC_DIAG=N").  Timing only.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
python3 scripts/experiments/time_cfgA.py
for d in 6 7 8; do
  FFTCONV_LIB=$PWD/fft_conv_pytorch_amd/libfftconv_diag$d.so python3 scripts/experiments/time_cfgA.py
done
python3 scripts/experiments/time_cfgA.py
