#!/bin/bash
# Submit one command to the GPU box; when no slot is free (gpurun exit 3: nothing ran, nothing charged) wait and
# submit again.  Only the SUBMISSION is retried -- a command that ran and failed is never re-run.
# Usage: gpu_submit.sh <timeout-seconds> '<command>'
TMO="$1"; shift
for attempt in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$TMO" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  if ! grep -q '"status": "transient"' /root/repo/gpurun_out/.last_call.json 2>/dev/null; then exit $rc; fi
  echo "[gpu_submit] no slot (attempt $attempt), waiting 45 s"
  sleep 45
done
exit 3
