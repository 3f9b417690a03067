set -x
mkdir -p gpurun_out/r02d
Q=learning-from-sparse-demonstrations_amd/csrc/build
timeout 400 python bench.py --no-cpu-baseline > gpurun_out/r02d/bench_layout_prefetch.json 2> gpurun_out/r02d/bench.err
timeout 400 python bench.py --no-cpu-baseline --library $Q/tune_776f4ece1b80744d_nopf.so > gpurun_out/r02d/bench_layout_noprefetch.json 2>> gpurun_out/r02d/bench.err
timeout 400 python bench.py --no-cpu-baseline > gpurun_out/r02d/bench_layout_prefetch2.json 2>> gpurun_out/r02d/bench.err
timeout 2400 python -m pytest tests -m gpu -q --durations=10 > gpurun_out/r02d/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r02d/pytest.log
tail -15 gpurun_out/r02d/pytest.log
for f in gpurun_out/r02d/bench_*.json; do echo $f; python -c "
import json,sys; d=json.load(open('$f')); print(d['value'], d['config']['kernel_ms'], d['config']['oc_iters_mean'])"; done
