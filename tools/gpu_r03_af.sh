#!/bin/bash
OUT=gpurun_out/r03af
mkdir -p $OUT
rm -f $OUT/parity_floors.jsonl
LFSD_PARITY_REPORT=$PWD/$OUT/parity_floors.jsonl timeout 900 python3 -m pytest tests -m gpu -q -k "rocket" > $OUT/pytest_gpu.txt 2>&1
tail -5 $OUT/pytest_gpu.txt; grep "rocket lean" $OUT/parity_floors.jsonl
