#!/bin/bash
# round 3, GPU session T: generic backward sweep (wide kernel, lock-step kernels without MFMA) -- double-buffered LDS rows in fp32
# (LFSD_BW_ROWBUF32), one stage of look-ahead on its global loads (LFSD_BW_PREFETCH_GEN), 3 stages in the wide costate sweep (LFSD_CS_AHEAD)
OUT=gpurun_out/r03t
mkdir -p $OUT
python3 tools/model_ab.py run rocket 100 1024 f32 product norowbuf r3pre > $OUT/rocket_ab.txt 2>&1
python3 tools/wide_clock.py run rocket 100 1024 f32 > $OUT/wide_clock_rocket.txt 2>&1
python3 tools/model_ab.py run robotarm 50 1024 f32 product norowbuf r3pre > $OUT/robotarm_ab.txt 2>&1
python3 tools/model_ab.py run pendulum 50 4096 f32 product > $OUT/pendulum.txt 2>&1
grep -v amdgpu $OUT/rocket_ab.txt $OUT/robotarm_ab.txt $OUT/pendulum.txt; grep -v amdgpu $OUT/wide_clock_rocket.txt | tail -7
