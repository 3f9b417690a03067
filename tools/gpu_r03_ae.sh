#!/bin/bash
# round 3, GPU session AE: final build at the node batch of configs[3] (32768 trajectories on one GPU), fp32 and fp64
OUT=gpurun_out/r03ae
mkdir -p $OUT
python3 bench.py --steps 5 --warmup 2 --batch 32768 --no-cpu-baseline > $OUT/bench_f32_32768.json 2> $OUT/err32.txt
python3 bench.py --steps 3 --warmup 1 --batch 32768 --dtype f64 --no-cpu-baseline > $OUT/bench_f64_32768.json 2> $OUT/err64.txt
python3 -c "
import json
for n in ('f32','f64'):
    o=json.load(open('$OUT/bench_%s_32768.json' % n)); print(n, round(o['value']), o['ms_per_step'], o['config']['kernel_ms'], o['config']['oc_status_hist'])"
