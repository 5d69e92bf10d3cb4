#!/bin/bash
# Bit-for-bit comparison of the producer / consumer backward (GSR_BWD_PC=1) with the barrier form (GSR_BWD_PC=0) on the GPU box
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
for cfg in "4 100000 800 0" "1 100000 800 0" "2 20000 400 1" "8 100000 800 0" "3 5000 200 0"; do
  set -- $cfg
  FROZEN=$4 GSR_BWD_PC=0 timeout 120 python $R/tools/r05_pc_check.py /tmp/a.npz $1 $2 $3 > /dev/null 2>$O/pc_check_err_a.txt || { echo "FAIL run A $cfg"; tail -5 $O/pc_check_err_a.txt; continue; }
  FROZEN=$4 GSR_BWD_PC=1 timeout 120 python $R/tools/r05_pc_check.py /tmp/b.npz $1 $2 $3 > /dev/null 2>$O/pc_check_err_b.txt || { echo "FAIL run B $cfg"; tail -5 $O/pc_check_err_b.txt; continue; }
  python - <<PY
import numpy as np
a, b = np.load("/tmp/a.npz"), np.load("/tmp/b.npz")
bad = [k for k in a.files if not np.array_equal(a[k], b[k])]
print("V P S frozen = $cfg:", "BIT-IDENTICAL" if not bad else "DIFFERS in %s, max abs %s" % (bad, [float(np.abs(a[k]-b[k]).max()) for k in bad]))
PY
done
