#!/bin/bash
# round 3, GPU session Y: phase clocks of the wide OC kernel on the robot arm (n_grid 50, 1024 seeds, fp32); smoke()
OUT=gpurun_out/r03y
mkdir -p $OUT
python3 tools/wide_clock.py run robotarm 50 1024 f32 > $OUT/wide_clock_robotarm.txt 2>&1
grep -v amdgpu $OUT/wide_clock_robotarm.txt | tail -8
python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -3 $OUT/smoke.txt
