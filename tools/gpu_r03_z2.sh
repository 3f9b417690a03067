#!/bin/bash
OUT=gpurun_out/r03z
mkdir -p $OUT
python3 tools/bw_clock.py bwclock f64 > $OUT/bw_clock_f64.txt 2>&1
grep -v amdgpu $OUT/bw_clock_f64.txt | tail -6
