#!/bin/bash
# round 4: the full GPU suite, the bench lines and the rocprofv3 profile set of the round (outputs under gpurun_out/r04p, copied to profiles/r04_*)
O=gpurun_out/r04p; mkdir -p $O
export GSR_ROW_MARGINS_LOG=$PWD/$O/row_margins.log; rm -f $GSR_ROW_MARGINS_LOG
( time GSR_TEST_VERBOSE=1 python -m pytest tests -m gpu -x -q -s --durations=8 ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --config 5 --steps 20 --warmup 3 > $O/bench_cfg5.json 2>> $O/bench.err
python bench.py --config 5 --with-rollout --steps 30 > $O/bench_cfg5_episode.json 2>> $O/bench.err
GSR_BENCH_SINGLE_DEVICE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 5 --warmup 2 > $O/bench_sd2.json 2>> $O/bench.err
bash tools/view_sweep.sh > $O/view_sweep.txt 2>&1
python tools/dropin_getloss_profile.py t0 > $O/dropin_profile_t0.txt 2>&1
python tools/dropin_getloss_profile.py > $O/dropin_profile_t1.txt 2>&1
for V in 4 8; do
  bash tools/prof_round.sh $V > $O/prof_round_v$V.log 2>&1
  for f in kernel_stats_v$V.txt pmc_traffic_v$V.json sq_counters_v$V.json sq_render_v$V.txt; do cp gpurun_out/$f $O/ 2>/dev/null; done
done
cp gpurun_out/marker_ranges.txt $O/ 2>/dev/null
bash tools/view_timeline.sh 1 > $O/timeline_v1.txt 2>&1
head -3 $O/view_sweep.txt; head -14 $O/kernel_stats_v4.txt
