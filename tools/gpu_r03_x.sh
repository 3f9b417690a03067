#!/bin/bash
# round 3, GPU session X: the final evidence set of the round (tools/gpu_profile.sh r03x) + the full -m gpu tier with parity floors
bash tools/gpu_profile.sh r03x
OUT=gpurun_out/r03x
rm -f $OUT/parity_floors.jsonl
LFSD_PARITY_REPORT=$PWD/$OUT/parity_floors.jsonl timeout 2700 python3 -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1
tail -4 $OUT/pytest_gpu.txt
