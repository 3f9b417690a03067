#!/bin/bash
OUT=gpurun_out/r03ac
mkdir -p $OUT
python3 tools/oc_clock64.py f64 > $OUT/oc_clock64.txt 2>&1
grep -v amdgpu $OUT/oc_clock64.txt | tail -14
