#!/bin/bash
# plain batch-sharing kernel: 8-byte loads (+1) / stores (+2) with v_permlane16_swap against 4-byte ones (FC_PAIR=0),
# with the planner's V = 513 (odd tile starts: half of the 8-byte accesses unaligned) and V = 512 (FFTCONV_VEVEN=1)
cd $GRAFT_REPO_ROOT/fft_conv_pytorch_amd/csrc
for pv in 0 1 3; do
  touch conv1d_pers.hpp
  make -j16 EXTRA="-fno-slp-vectorize -DFC_PAIR=$pv" > /dev/null 2>&1 || exit 1
  for ve in 0 1; do
    echo "== FC_PAIR=$pv VEVEN=$ve"
    (cd $GRAFT_REPO_ROOT && FFTCONV_VEVEN=$ve timeout -k 10 120 python3 scripts/variant_check.py --tag "pair${pv}_veven$ve" 2>/dev/null | grep tag) || exit 1
  done
done
