#!/bin/bash
cd $GRAFT_REPO_ROOT
for V in 2 4; do
  echo "== views $V"
  bash tools/ab_env.sh "--views $V --no-optimizer" "GSR_BWD_SMALL_BATCH_TILES=100000" "GSR_BWD_SMALL_BATCH_TILES=0" 2>&1 | cut -c1-200
done
