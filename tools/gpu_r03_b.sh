#!/bin/bash
# round 3, GPU session B: the whole -m gpu tier with the parity report (measured vs asserted of every comparison), arm divergence data
OUT=gpurun_out/r03b
mkdir -p $OUT
rm -f $OUT/parity_floors.jsonl
LFSD_PARITY_REPORT=$PWD/$OUT/parity_floors.jsonl timeout 2400 python3 -m pytest tests -m gpu -q -x --durations=15 > $OUT/pytest_gpu.txt 2>&1
tail -40 $OUT/pytest_gpu.txt
timeout 600 python3 tools/arm_j0.py > $OUT/arm_j0.txt 2>&1
tail -40 $OUT/arm_j0.txt
