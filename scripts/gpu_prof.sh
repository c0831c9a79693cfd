#!/bin/bash
# rocprofv3 session for bench.py: kernel trace + stats, then PMC passes (each in its own run,
# never combined with other trace domains).  Usage: gpu_prof.sh <tag> [bench args...]
set -u
ROOTDIR="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOTDIR/gpurun_out"
TAG="${1:-prof}"; shift || true
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
run() {  # name, timeout, cmd...
  local name=$1 tmo=$2; shift 2
  echo "=== $name"
  timeout -k 10 "$tmo" "$@" > "$OUT/$name.log" 2>&1
  local rc=$?
  echo "rc=$rc"; tail -n 4 "$OUT/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping"; exit 1; fi
}
BENCH="python3 $ROOTDIR/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-end-to-end $*"
run ${TAG}_trace 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_trace" -- $BENCH
i=0
for PMC in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  run ${TAG}_pmc$i 300 rocprofv3 --pmc $PMC --output-format csv -d "$OUT/${TAG}_pmc$i" -- $BENCH
done
cd "$ROOTDIR"
python3 scripts/summarize_prof.py "$OUT" "$TAG" | tee "$OUT/${TAG}_summary.txt"
exit 0
