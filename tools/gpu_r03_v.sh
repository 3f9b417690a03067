#!/bin/bash
# round 3, GPU session V: load balance of the headline kernels (per-trajectory work vs wavefront / SIMD)
OUT=gpurun_out/r03v
mkdir -p $OUT
python3 tools/aux_balance.py 6 > $OUT/aux_balance.txt 2>&1
grep -v amdgpu $OUT/aux_balance.txt | tail -12
