#!/bin/bash
# round 3, GPU session U: product build after the generic-backward changes: other configurations, fp32 + fp64 bench lines, full -m gpu tier
OUT=gpurun_out/r03u
mkdir -p $OUT
python3 tools/other_configs.py > $OUT/other_configs.txt 2>&1
grep -v amdgpu $OUT/other_configs.txt | tail -6
python3 bench.py --steps 5 --warmup 2 --dtype f64 --no-cpu-baseline > $OUT/bench_f64.json 2> $OUT/bench_f64.err
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_f32.json 2> $OUT/bench_f32.err
python3 -c "
import json
for n in ('f64','f32'):
    o=json.load(open('$OUT/bench_%s.json' % n)); print(n, round(o['value']), o['ms_per_step'], o['config']['kernel_ms'], o['config']['oc_status_hist'], o['roofline']['valu_frac'], o['roofline']['mfma_useful_tflops'])"
rm -f $OUT/parity_floors.jsonl
LFSD_PARITY_REPORT=$PWD/$OUT/parity_floors.jsonl timeout 2700 python3 -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1
tail -4 $OUT/pytest_gpu.txt
