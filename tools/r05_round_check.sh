# Round check on the GPU box (tools/r05_round_check.sh): the whole GPU suite, smoke, the driver bench line, cfg5 frame + 60-frame episode, the 2-rank smoke, the profile set of tools/prof_round.sh at four views -> gpurun_out/r05_round_check/
O=gpurun_out/r05_round_check; mkdir -p $O
export GSR_ROW_MARGINS_LOG=$PWD/$O/row_margins.log
( time python -m pytest tests -m gpu -x -q --durations=10 ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -16 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --config 5 > $O/bench_cfg5.json 2> $O/bench_cfg5.err
python bench.py --config 5 --with-rollout --steps 60 > $O/bench_cfg5_ep60.json 2> $O/bench_cfg5_ep60.err
GSR_BENCH_SINGLE_DEVICE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 > $O/bench_sd2.json 2> $O/bench_sd2.err
bash tools/prof_round.sh 4 > $O/prof_round4.log 2>&1
mkdir -p $O/prof4; cp gpurun_out/kernel_stats_v4.txt gpurun_out/pmc_traffic*.json gpurun_out/sq_counters*.json gpurun_out/sq_render*.txt gpurun_out/marker_ranges.txt $O/prof4/ 2>/dev/null
head -c 400 $O/bench.json; echo; head -12 gpurun_out/kernel_stats_v4.txt
