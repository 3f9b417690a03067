#!/bin/bash
# round 3, GPU session AD: closed-loop control law with its operands staged in LDS one interval ahead (LFSD_CTL_STAGE) vs read in place
OUT=gpurun_out/r03ad
mkdir -p $OUT
python3 tools/ab_variants.py run base nostage base nostage --steps 20 > $OUT/ab_f32.txt 2>&1
python3 tools/ab_variants.py run base nostage --steps 5 -- --dtype f64 > $OUT/ab_f64.txt 2>&1
python3 tools/model_ab.py run rocket 100 1024 f32 product nostage > $OUT/rocket_ab.txt 2>&1
python3 tools/model_ab.py run robotarm 50 1024 f32 product nostage > $OUT/robotarm_ab.txt 2>&1
grep -v amdgpu $OUT/ab_f32.txt $OUT/ab_f64.txt $OUT/rocket_ab.txt $OUT/robotarm_ab.txt
