set -x
mkdir -p gpurun_out/r02f
Q=learning-from-sparse-demonstrations_amd/csrc/build
./tools/probes/mfma_probe 2>&1 | head -4 > gpurun_out/r02f/mfma_probe.txt; cat gpurun_out/r02f/mfma_probe.txt
timeout 400 python bench.py --no-cpu-baseline > gpurun_out/r02f/bench_mf1.json 2> gpurun_out/r02f/bench.err
timeout 400 python bench.py --no-cpu-baseline --library $Q/tune_quadrotor_mf0.so > gpurun_out/r02f/bench_mf0.json 2>> gpurun_out/r02f/bench.err
timeout 400 python bench.py --no-cpu-baseline > gpurun_out/r02f/bench_mf1b.json 2>> gpurun_out/r02f/bench.err
for f in gpurun_out/r02f/bench_*.json; do echo $f; python -c "
import json,sys; d=json.load(open('$f')); print(d['value'], d['config']['kernel_ms'], d['config']['oc_iters_mean'], d['config']['oc_status_hist'])"; done
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r02f/pytest.log 2>&1; tail -5 gpurun_out/r02f/pytest.log
