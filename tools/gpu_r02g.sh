set -x
mkdir -p gpurun_out/r02g
timeout 900 python tools/wide_vs_lockstep.py > gpurun_out/r02g/wide_vs_lockstep.txt 2>&1; cat gpurun_out/r02g/wide_vs_lockstep.txt
timeout 2400 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r02g/pytest.log 2>&1; tail -14 gpurun_out/r02g/pytest.log
timeout 600 python tools/other_configs.py > gpurun_out/r02g/other.txt 2>&1; cat gpurun_out/r02g/other.txt
