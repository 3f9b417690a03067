#!/bin/bash
# round 3, GPU session O: fp64 lean kernel with / without the scheduling fences of the fp64 instantiations; phase clocks
OUT=gpurun_out/r03o
mkdir -p $OUT
python3 tools/ab_variants.py run base pin64only nofence64 --steps 5 -- --dtype f64 > $OUT/ab_f64.txt 2>&1
python3 tools/ab_variants.py run base nofence64 --steps 20 > $OUT/ab_f32.txt 2>&1
python3 tools/oc_clock64.py f64 > $OUT/oc_clock64.txt 2>&1
grep -v amdgpu $OUT/ab_f64.txt $OUT/ab_f32.txt; tail -8 $OUT/oc_clock64.txt
