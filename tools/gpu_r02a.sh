set -x
mkdir -p gpurun_out/r02a
python -m pytest tests -m gpu -x -q > gpurun_out/r02a/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r02a/pytest.log
python bench.py > gpurun_out/r02a/bench.json 2> gpurun_out/r02a/bench.err; echo "bench rc $?"
python tools/arm_steps.py > gpurun_out/r02a/arm_steps.txt 2>&1
python tools/diag_lockstep.py > gpurun_out/r02a/lockstep.txt 2>&1
python tools/other_configs.py > gpurun_out/r02a/other.txt 2>&1
tail -3 gpurun_out/r02a/pytest.log; cat gpurun_out/r02a/bench.json
