# round 5, second session, first GPU call: state check after the container restore + rollout trace
O=gpurun_out/s2a; mkdir -p $O; R=$PWD
( time python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --config 5 --with-rollout > $O/bench_cfg5_ep.json 2> $O/bench_cfg5_ep.err
python tools/rollout_graph_loop.py > $O/graph_loop.txt 2>&1
python tools/rollout_episode_profile.py > $O/episode_profile.txt 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_roll -o run -- python $R/tools/rollout_graph_loop.py > $R/$O/prof_roll.log 2>&1
cd $R; tail -3 $O/pytest.log; cat $O/graph_loop.txt; head -c 600 $O/bench.json
