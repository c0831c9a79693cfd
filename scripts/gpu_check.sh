#!/bin/bash
# One GPU-box session: parity tests -> bench -> rocprofv3 kernel trace.  Stops at the first step
# that times out (a hung kernel must not be followed by more GPU work).
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
OUT="$(pwd)/gpurun_out"
mkdir -p $OUT
SEL="${1:-}"
step() {  # name, timeout, command...
  local name=$1 tmo=$2; shift 2
  echo "=== $name" | tee -a $OUT/session.log
  timeout -k 10 "$tmo" "$@" > "$OUT/$name.log" 2>&1
  local rc=$?
  echo "rc=$rc" | tee -a $OUT/session.log
  tail -n 15 "$OUT/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a $OUT/session.log; exit 1; fi
  return $rc
}
: > $OUT/session.log
if [ -n "$SEL" ]; then
  step tests 900 python -m pytest tests -q -m gpu --maxfail=10 -k "$SEL" -s
else
  step tests 900 python -m pytest tests -q -m gpu --maxfail=10 -s
fi
step bench 400 python bench.py --steps 400 --warmup 40
export TMPDIR=/tmp
ROOTDIR=$(pwd)
cd /tmp
step rocprof 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- python3 "$ROOTDIR/bench.py" --steps 200 --warmup 20 --no-cpu-baseline
cd "$ROOTDIR"
find $OUT/prof -name "*stats*.csv" | head -5 | while read f; do echo "--- $f"; head -12 "$f"; done
exit 0
