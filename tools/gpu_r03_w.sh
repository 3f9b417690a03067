#!/bin/bash
# round 3, GPU session W: rows of the wide backward sweep's dense products split over the four lane quarters (LFSD_BW_QSPLIT), rocket
OUT=gpurun_out/r03w
mkdir -p $OUT
python3 tools/model_ab.py run rocket 100 1024 f32 product noqs > $OUT/rocket_ab.txt 2>&1
python3 tools/wide_clock.py run rocket 100 1024 f32 > $OUT/wide_clock_rocket.txt 2>&1
grep -v amdgpu $OUT/rocket_ab.txt; grep -v amdgpu $OUT/wide_clock_rocket.txt | tail -4
