#!/bin/bash
# round 3, GPU session Z: one-pass fp64 backward sweep in the structural layout (LFSD_FP64_SC) vs two passes of the generic sweep
OUT=gpurun_out/r03z
mkdir -p $OUT
python3 tools/ab_variants.py run base nosc64 --steps 5 -- --dtype f64 > $OUT/ab_f64.txt 2>&1
python3 tools/oc_clock64.py f64 > $OUT/oc_clock64.txt 2>&1
grep -v amdgpu $OUT/ab_f64.txt; tail -3 $OUT/oc_clock64.txt
timeout 1500 python3 -m pytest tests -m gpu -q -k "quadrotor or headline or float64 or f64" > $OUT/pytest_gpu.txt 2>&1; python3 tools/bw_clock.py bwclock f64 > $OUT/bw_clock_f64.txt 2>&1; grep -v amdgpu $OUT/bw_clock_f64.txt | tail -2
tail -4 $OUT/pytest_gpu.txt
