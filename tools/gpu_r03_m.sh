#!/bin/bash
# round 3, GPU session M: shared-mode line (mesh continuation with the learner's all-zero initial guesses), rocprofv3 kernel stats
# of the OTHER configurations (robot arm, rocket, pendulum, quadrotor 32768), two-rank bench, full -m gpu tier
OUT=gpurun_out/r03m
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python3 bench.py --steps 10 --warmup 2 --mode shared --no-cpu-baseline > $OUT/bench_shared.json 2> $OUT/bench_shared.err
python3 -c "
import json; o=json.load(open('$OUT/bench_shared.json')); print('shared', round(o['value']), o['ms_per_step'], o['config']['kernel_ms'], o['config']['oc_status_hist'], o['config']['n_unconverged_last_step'])"
rocprofv3 --kernel-trace --stats -d $OUT/prof_other -o other --output-format csv -- python3 tools/other_configs.py > $OUT/other_configs.txt 2> $OUT/other_prof.err
cp $(find $OUT/prof_other -name "*kernel_stats.csv") $OUT/other_configs_kernel_stats.csv
rm -rf $OUT/prof_other
grep -v amdgpu $OUT/other_configs.txt | tail -6
rm -f $OUT/parity_floors.jsonl
LFSD_PARITY_REPORT=$PWD/$OUT/parity_floors.jsonl timeout 2700 python3 -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1
tail -4 $OUT/pytest_gpu.txt
