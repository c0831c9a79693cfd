for pl in 1 2 0; do
  cd /tmp && export TMPDIR=/tmp
  FFTCONV_PLANES=$pl timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2_planes3_${pl}_trace -- python3 $GRAFT_REPO_ROOT/bench.py --config cfgC --steps 40 --warmup 5 --no-cpu-baseline --no-end-to-end > $GRAFT_REPO_ROOT/gpurun_out/r2_planes3_${pl}_trace.log 2>&1
  cd $GRAFT_REPO_ROOT
  echo "== FFTCONV_PLANES=$pl"; python3 scripts/summarize_prof.py gpurun_out r2_planes3_${pl} | grep "fc::" | cut -c1-110
  grep "\"metric\"" gpurun_out/r2_planes3_${pl}_trace.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', round(d['roofline']['kernel_us'],1), 'us')"
done
