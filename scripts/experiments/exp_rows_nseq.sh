#!/bin/bash
# rows passes of the N-d path: pair-sequences per workgroup = NSEQ_C / FC_ROWS_DIV (2 = shipped: 16 rows, 128-byte runs of the
# transposed stores / loads; 1: 32 rows, 256-byte runs).  Rebuilds on the GPU box.
cd $GRAFT_REPO_ROOT/fft_conv_pytorch_amd/csrc
for dv in 1 2; do
  touch tile_inst.hip
  make -j16 EXTRA="-fno-slp-vectorize -DFC_ROWS_DIV=$dv" > /dev/null 2>&1 || exit 1
  for cfg in cfgB cfgC; do
    echo "== FC_ROWS_DIV=$dv $cfg"
    (cd /tmp && TMPDIR=/tmp timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2_rows${dv}_${cfg}_trace -- python3 $GRAFT_REPO_ROOT/bench.py --config $cfg --steps 40 --warmup 5 --no-cpu-baseline --no-end-to-end > $GRAFT_REPO_ROOT/gpurun_out/r2_rows${dv}_${cfg}.log 2>&1) || exit 1
    python3 $GRAFT_REPO_ROOT/scripts/summarize_prof.py $GRAFT_REPO_ROOT/gpurun_out r2_rows${dv}_${cfg} | grep "fc::" | cut -c1-100
    grep '"metric"' $GRAFT_REPO_ROOT/gpurun_out/r2_rows${dv}_${cfg}.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', round(d['roofline']['kernel_us'],1), 'us')"
  done
done
