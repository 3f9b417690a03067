#!/bin/bash
# round 3, GPU session F: rocket iteration trace; the whole -m gpu tier on the product build with the parity report; bench
OUT=gpurun_out/r03f
mkdir -p $OUT
timeout 900 python3 tools/oc_trace.py run rocket 100 1024 f32 2 > $OUT/rocket_trace.txt 2>&1
grep -v "^wide it" $OUT/rocket_trace.txt | tail -8
python3 bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json
rm -f $OUT/parity_floors.jsonl
LFSD_PARITY_REPORT=$PWD/$OUT/parity_floors.jsonl timeout 2400 python3 -m pytest tests -m gpu -q --durations=8 > $OUT/pytest_gpu.txt 2>&1
tail -25 $OUT/pytest_gpu.txt
