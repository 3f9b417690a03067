#!/bin/bash
# round 3, GPU session AA: fp32 structural backward sweep with V_x / costate read from LDS (LFSD_SC_VX_LDS=1) vs carried in registers
OUT=gpurun_out/r03aa
mkdir -p $OUT
python3 tools/ab_variants.py run base vxlds base vxlds --steps 20 > $OUT/ab_f32.txt 2>&1
grep -v amdgpu $OUT/ab_f32.txt
