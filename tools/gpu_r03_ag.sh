#!/bin/bash
# round 3, GPU session AG: the round's final build -- fp64 / fp32 bench lines, smoke(), full -m gpu tier with parity floors
OUT=gpurun_out/r03ag
mkdir -p $OUT
python3 bench.py --steps 10 --warmup 2 --dtype f64 --no-cpu-baseline > $OUT/bench_f64.json 2> $OUT/bench_f64.err
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_f32.json 2> $OUT/bench_f32.err
python3 -c "
import json
for n in ('f64','f32'):
    o=json.load(open('$OUT/bench_%s.json' % n)); print(n, round(o['value']), o['ms_per_step'], o['config']['kernel_ms'], o['config']['oc_status_hist'], o['roofline']['valu_frac'])"
python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
rm -f $OUT/parity_floors.jsonl
LFSD_PARITY_REPORT=$PWD/$OUT/parity_floors.jsonl timeout 2700 python3 -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1
tail -4 $OUT/pytest_gpu.txt
