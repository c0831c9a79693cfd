#!/bin/bash
# per-kernel times of the many-channel pipeline on one shape of scripts/dense_check.py (argument: shape index)
IDX=${1:-0}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2_dense${IDX}_trace -- python3 $GRAFT_REPO_ROOT/scripts/dense_check.py $IDX > $GRAFT_REPO_ROOT/gpurun_out/r2_dense${IDX}_trace.log 2>&1 || exit 1
cd $GRAFT_REPO_ROOT
python3 scripts/summarize_prof.py gpurun_out r2_dense${IDX} | cut -c1-150
