#!/bin/bash
# Round 6: the profile set the docs and bench.py's replay cite, on the final tree: kernel stats + PMC traffic + SQ counters (4 views = BASELINE's
# configuration), view sweep, one-view timeline, the default bench line, configs[4]'s frame loop, the N > 1 code path on one device (gloo, not a perf number)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O/r06_final
bash $R/tools/prof_round.sh 4 > $O/r06_final/prof_round4.log 2>&1
bash $R/tools/view_sweep.sh > /dev/null 2>&1
bash $R/tools/view_timeline.sh 1 > /dev/null 2>&1
for f in kernel_stats_v4.txt pmc_traffic_v4.json sq_counters_v4.json sq_render_v4.txt view_sweep.txt timeline_v1.txt marker_ranges.txt; do cp $O/$f $O/r06_final/ 2>/dev/null; done
# the default line replays profiles/r06_*: give it this run's
cp $O/r06_final/kernel_stats_v4.txt $R/profiles/r06_kernel_stats_v4.txt; cp $O/r06_final/pmc_traffic_v4.json $R/profiles/r06_pmc_traffic_v4.json; cp $O/r06_final/sq_counters_v4.json $R/profiles/r06_sq_counters_v4.json
python $R/bench.py > $O/r06_final/bench.json 2> $O/r06_final/bench.err
python $R/bench.py --config 5 --steps 12 > $O/r06_final/bench_cfg5.json 2> $O/r06_final/bench_cfg5.err
GSR_BENCH_SINGLE_DEVICE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 $R/bench.py --gpus 2 --steps 5 --warmup 2 > $O/r06_final/bench_sd2.json 2> $O/r06_final/bench_sd2.err
tail -c 300 $O/r06_final/bench.json; echo; cat $O/r06_final/view_sweep.txt; head -12 $O/r06_final/kernel_stats_v4.txt
