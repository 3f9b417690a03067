#!/bin/bash
# Evidence for profiles/: rocprofv3 kernel stats of `bench.py`, then the PMC passes (separate runs, --kernel-trace only
# beside --pmc).  usage (on the GPU box, via gpurun): bash tools/gpu_profile.sh <tag>
set -x
TAG=${1:-r02}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd -
rocprofv3 -L > $OUT/counters_list.txt 2>&1
grep -i -o "SQ_[A-Z_]*MFMA[A-Z_]*" $OUT/counters_list.txt | sort -u > $OUT/mfma_counter_names.txt; cat $OUT/mfma_counter_names.txt
python3 bench.py --steps 10 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT/prof -o $TAG --output-format csv -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_prof.json 2> $OUT/prof.err
B="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o fetch --output-format csv -- $B > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o write --output-format csv -- $B > $OUT/pmc_write.json 2> $OUT/pmc_write.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -d $OUT/pmc_valu -o valu --output-format csv -- $B > $OUT/pmc_valu.json 2> $OUT/pmc_valu.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES -d $OUT/pmc_mfma -o mfma --output-format csv -- $B > $OUT/pmc_mfma.json 2> $OUT/pmc_mfma.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_ANY -d $OUT/pmc_mfma2 -o mfma2 --output-format csv -- $B > $OUT/pmc_mfma2.json 2> $OUT/pmc_mfma2.err
find $OUT -name "*counter_collection.csv" -o -name "*kernel_stats.csv" | head -20
python3 tools/hbm_traffic.py $(find $OUT/pmc_fetch -name "*counter_collection.csv") $(find $OUT/pmc_write -name "*counter_collection.csv") $OUT/hbm_traffic.json
python3 tools/issue_counters.py $OUT/issue_counters.json $(find $OUT/pmc_valu -name "*counter_collection.csv") $(find $OUT/pmc_mfma -name "*counter_collection.csv")
tail -3 $OUT/pmc_mfma.err $OUT/pmc_mfma2.err
cat $OUT/bench.json
