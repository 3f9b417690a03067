#!/bin/bash
# round 3, GPU session E: later exits from the coarse grid; iteration trace of the rocket's slowest solves (n_grid 100, 1024 seeds)
OUT=gpurun_out/r03e
mkdir -p $OUT
python3 tools/ab_variants.py run base cs1e3 cs1e4 cs0 --steps 20 --batch 4096 > $OUT/ab_variants.txt 2>&1
cat $OUT/ab_variants.txt
timeout 900 python3 tools/oc_trace.py run rocket 100 1024 f32 2 > $OUT/rocket_trace.txt 2>&1
grep -v "^wide it" $OUT/rocket_trace.txt | tail -12
