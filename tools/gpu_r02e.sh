set -x
mkdir -p gpurun_out/r02e
Q=learning-from-sparse-demonstrations_amd/csrc/build
./tools/probes/mfma_probe > gpurun_out/r02e/mfma_probe.txt 2>&1
head -3 gpurun_out/r02e/mfma_probe.txt
timeout 900 python -m pytest tests -m gpu -q -x -k "quadrotor or rocket" > gpurun_out/r02e/pytest_quad.log 2>&1; tail -5 gpurun_out/r02e/pytest_quad.log
timeout 400 python bench.py --no-cpu-baseline > gpurun_out/r02e/bench_mf1.json 2> gpurun_out/r02e/bench.err
timeout 400 python bench.py --no-cpu-baseline --library $Q/tune_*_mf0.so > gpurun_out/r02e/bench_mf0.json 2>> gpurun_out/r02e/bench.err
timeout 400 python bench.py --no-cpu-baseline > gpurun_out/r02e/bench_mf1b.json 2>> gpurun_out/r02e/bench.err
for f in gpurun_out/r02e/bench_*.json; do echo $f; python -c "
import json,sys; d=json.load(open('$f')); print(d['value'], d['config']['kernel_ms'], d['config']['oc_iters_mean'], d['config']['oc_status_hist'])"; done
