#!/bin/bash
# round 3, GPU session D: how to leave the coarse grid / when (A/B), then the quadrotor parity tests on the product build
OUT=gpurun_out/r03d
mkdir -p $OUT
python3 tools/ab_variants.py run base relinhard cs1e2 cs1e3 cs3 nocoarse --steps 20 --batch 4096 > $OUT/ab_variants.txt 2>&1
cat $OUT/ab_variants.txt
LFSD_PARITY_REPORT=$PWD/$OUT/parity_floors.jsonl timeout 1500 python3 -m pytest tests -m gpu -q -k "quadrotor or headline or bench or full_size_properties_quad" > $OUT/pytest_gpu.txt 2>&1
tail -15 $OUT/pytest_gpu.txt
