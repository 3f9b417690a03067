set -x
mkdir -p gpurun_out/r02c
timeout 2400 python -m pytest tests -m gpu -q --durations=25 > gpurun_out/r02c/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r02c/pytest.log
timeout 300 python bench.py --gpus 2 --backend gloo --steps 3 --warmup 1 --batch 1024 > gpurun_out/r02c/bench_2rank_gloo.json 2> gpurun_out/r02c/bench_2rank.err; echo "2-rank rc $?"
timeout 300 python bench.py --mode shared --steps 5 --warmup 2 > gpurun_out/r02c/bench_shared_n1.json 2> gpurun_out/r02c/bench_shared.err
timeout 400 python bench.py > gpurun_out/r02c/bench.json 2> gpurun_out/r02c/bench.err
tail -40 gpurun_out/r02c/pytest.log; cat gpurun_out/r02c/bench_2rank_gloo.json; tail -3 gpurun_out/r02c/bench_2rank.err
