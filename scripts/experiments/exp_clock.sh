#!/bin/bash
# In-kernel clock and item lifetime against the number of busy CUs (batch 32 = all 256, 16 = 126, 8 = 63 workgroups)
cd $GRAFT_REPO_ROOT
for b in 32 16 8 4; do
  echo "== batch $b"
  timeout -k 10 120 python3 scripts/phase_profile.py --batch $b | grep -E "items=|clock|lifetime|mix done|passA done|passB done" || exit 1
done
