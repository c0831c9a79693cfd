#!/bin/bash
# channel pairs per workgroup of dense_fwd / dense_inv: 16 (512 threads, one workgroup per CU, 256-byte runs) against
# 8 (256 threads, two workgroups per CU, 128-byte runs).  Rebuilds on the GPU box.
cd $GRAFT_REPO_ROOT/fft_conv_pytorch_amd/csrc
for n in 8 16; do
  touch tile_inst.hip
  make -j16 EXTRA="-fno-slp-vectorize -DFC_DENSE_NSEQ=$n" > /dev/null 2>&1 || exit 1
  echo "== FC_DENSE_NSEQ=$n"
  for idx in 0 4; do
    (cd /tmp && TMPDIR=/tmp timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2_nseq${n}_${idx}_trace -- python3 $GRAFT_REPO_ROOT/scripts/dense_check.py $idx > /dev/null 2>&1) || exit 1
    python3 $GRAFT_REPO_ROOT/scripts/summarize_prof.py $GRAFT_REPO_ROOT/gpurun_out r2_nseq${n}_${idx} | grep "fc::dense" | cut -c1-100
  done
done
