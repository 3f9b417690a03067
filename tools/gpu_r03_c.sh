#!/bin/bash
# round 3, GPU session C: A/B of the structural-columns sweep, the mesh continuation and the Riccati MFMA experiment on the
# headline workload; the SQ counters of the MFMA experiment before / after; quadrotor parity tests
OUT=gpurun_out/r03c
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python3 tools/ab_variants.py run base nocoarse nostruct r02like ricmfma --steps 20 --batch 4096 > $OUT/ab_variants.txt 2>&1
cat $OUT/ab_variants.txt
V=$(ls learning-from-sparse-demonstrations_amd/csrc/build/ab_*_ricmfma.so | head -1)
for tag in base ricmfma; do
  LIB=""; [ $tag = ricmfma ] && LIB="--library $V"
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_ANY -d $OUT/pmc_$tag -o $tag --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline $LIB > $OUT/pmc_$tag.json 2> $OUT/pmc_$tag.err
  cp $(find $OUT/pmc_$tag -name "*counter_collection.csv") $OUT/pmc_mfma_aux_${tag}_counter_collection.csv
done
LFSD_PARITY_REPORT=$PWD/$OUT/parity_floors.jsonl timeout 1500 python3 -m pytest tests -m gpu -q -x -k "quadrotor or headline or bench or full_size_properties_quad" > $OUT/pytest_gpu.txt 2>&1
tail -15 $OUT/pytest_gpu.txt
