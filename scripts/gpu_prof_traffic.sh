#!/bin/bash
# rocprofv3 kernel stats + HBM traffic counters (FETCH_SIZE / WRITE_SIZE, each in its own pass, never combined with
# other trace domains) of one bench.py configuration.  Usage: gpu_prof_traffic.sh <tag> <bench args...>
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
  echo "rc=$rc"; tail -n 2 "$OUT/$name.log" | cut -c1-300
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping"; exit 1; fi
}
BENCH="python3 $ROOTDIR/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-end-to-end $*"
run ${TAG}_trace 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_trace" -- $BENCH
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  run ${TAG}_pmc$i 300 rocprofv3 --pmc $PMC --output-format csv -d "$OUT/${TAG}_pmc$i" -- $BENCH
done
cd "$ROOTDIR"
python3 scripts/summarize_prof.py "$OUT" "$TAG" | grep -v "at::native\|rocclr" | tee "$OUT/${TAG}_summary.txt"
exit 0
