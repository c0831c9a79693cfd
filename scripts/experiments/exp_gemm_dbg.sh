#!/bin/bash
# Which part of dense_gemm sets its time: normal build, build without the LDS reads of the k loop, build without MFMA
# (results of the two diagnostic builds are wrong on purpose).  Rebuilds the library on the GPU box, restores it at the end.
cd $GRAFT_REPO_ROOT/fft_conv_pytorch_amd/csrc
for dbg in 1 2 0; do
  touch dense1d.hpp
  make -j16 EXTRA="-fno-slp-vectorize -DFC_DENSE_DBG=$dbg" > /dev/null 2>&1 || exit 1
  echo "== FC_DENSE_DBG=$dbg"
  (cd /tmp && TMPDIR=/tmp timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2_gdbg${dbg}_trace -- python3 $GRAFT_REPO_ROOT/scripts/dense_check.py 0 > /dev/null 2>&1) || exit 1
  python3 $GRAFT_REPO_ROOT/scripts/summarize_prof.py $GRAFT_REPO_ROOT/gpurun_out r2_gdbg${dbg} | grep "fc::dense" | cut -c1-100
done
