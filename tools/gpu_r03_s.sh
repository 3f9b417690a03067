#!/bin/bash
# round 3, GPU session S: phase clocks of the wide OC kernel on the rocket (n_grid 100, 1024 seeds, fp32), slowest solves
OUT=gpurun_out/r03s
mkdir -p $OUT
python3 tools/wide_clock.py run rocket 100 1024 f32 > $OUT/wide_clock_rocket.txt 2>&1
grep -v amdgpu $OUT/wide_clock_rocket.txt | tail -14
