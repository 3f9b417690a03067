#!/bin/bash
# round 3, GPU session R: double-buffered row reads in the fp64 backward sweep (LFSD_FENCE64 7 vs 3); phase clocks; fp64 parity
OUT=gpurun_out/r03r
mkdir -p $OUT
python3 tools/ab_variants.py run base fence3 --steps 5 -- --dtype f64 > $OUT/ab_f64.txt 2>&1
python3 tools/oc_clock64.py f64 > $OUT/oc_clock64.txt 2>&1
grep -v amdgpu $OUT/ab_f64.txt; tail -3 $OUT/oc_clock64.txt
timeout 1500 python3 -m pytest tests -m gpu -q -k "quadrotor or headline or float64 or f64" > $OUT/pytest_gpu.txt 2>&1
tail -4 $OUT/pytest_gpu.txt
