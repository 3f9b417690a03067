#!/bin/bash
# round 3, GPU session Q: fp64 lean kernel on 16-lane groups (live columns) vs 32-lane groups; phase clocks; fp64 parity tests
OUT=gpurun_out/r03q
mkdir -p $OUT
python3 tools/ab_variants.py run base nopark64 --steps 5 -- --dtype f64 > $OUT/ab_f64.txt 2>&1
python3 tools/oc_clock64.py f64 > $OUT/oc_clock64.txt 2>&1
grep -v amdgpu $OUT/ab_f64.txt $OUT/ab_f64_32k.txt; tail -3 $OUT/oc_clock64.txt
timeout 1500 python3 -m pytest tests -m gpu -q -k "quadrotor or headline or float64 or f64" > $OUT/pytest_gpu.txt 2>&1
tail -4 $OUT/pytest_gpu.txt
