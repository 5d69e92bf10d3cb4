mkdir -p gpurun_out/r04a
export GSR_ROW_MARGINS_LOG=$PWD/gpurun_out/r04a/row_margins.log
( time python -m pytest tests -m gpu -x -q --durations=15 ) > gpurun_out/r04a/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04a/pytest.log
python bench.py > gpurun_out/r04a/bench.json 2> gpurun_out/r04a/bench.err
GSR_BENCH_SINGLE_DEVICE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/r04a/bench_sd2.json 2> gpurun_out/r04a/bench_sd2.err
tail -5 gpurun_out/r04a/pytest.log
