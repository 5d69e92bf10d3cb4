#!/bin/bash
# round 4: the single-view entry points on the tile-row binning: full GPU suite + A/B against the radix path (GSR_RADIX_BINNING=1) on the drop-in numbers
O=gpurun_out/r04g; mkdir -p $O
( python -m pytest tests -m gpu -x -q 2>&1 | tail -4 ) > $O/pytest.log 2>&1; cat $O/pytest.log
bash tools/ab_dropin.sh "GSR_RADIX_BINNING=1" "GSR_RADIX_BINNING=0" > $O/ab_dropin.txt 2>&1; cat $O/ab_dropin.txt
