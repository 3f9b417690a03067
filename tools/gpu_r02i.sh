set -x
mkdir -p gpurun_out/r02i
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02i/smoke.log 2>&1; tail -3 gpurun_out/r02i/smoke.log
timeout 2400 python -m pytest tests -m gpu -q --durations=6 > gpurun_out/r02i/pytest.log 2>&1; tail -12 gpurun_out/r02i/pytest.log
timeout 400 python bench.py > gpurun_out/r02i/bench.json 2> gpurun_out/r02i/bench.err; cat gpurun_out/r02i/bench.json
timeout 300 python bench.py --gpus 2 --backend gloo --steps 3 --warmup 1 --batch 2048 --no-cpu-baseline > gpurun_out/r02i/bench_2rank_gloo.json 2> gpurun_out/r02i/bench2.err; cut -c1-300 gpurun_out/r02i/bench_2rank_gloo.json
