#!/bin/bash
# Evidence for profiles/: bench line, rocprofv3 kernel stats of `bench.py`, then the PMC passes (separate runs, --kernel-trace
# only beside --pmc), folded per FULL-BATCH dispatch.  usage (on the GPU box, via gpurun): bash tools/gpu_profile.sh <tag>
TAG=${1:-r03}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python3 bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT/prof -o $TAG --output-format csv -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_prof.json 2> $OUT/prof.err
B="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline"
pass() {  # name, counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d $OUT/pmc_$name -o $name --output-format csv -- $B > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err
  cp $(find $OUT/pmc_$name -name "*counter_collection.csv") $OUT/pmc_${name}_counter_collection.csv
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass lane SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS
pass mfma SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
pass flops SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FLOPS_FP32 SQ_THREAD_CYCLES_VALU
cp $(find $OUT/prof -name "*kernel_stats.csv") $OUT/kernel_stats.csv
python3 tools/hbm_traffic.py $OUT/pmc_fetch_counter_collection.csv $OUT/pmc_write_counter_collection.csv $OUT/hbm_traffic.json
python3 tools/issue_counters.py $OUT/issue_counters.json $OUT/pmc_lane_counter_collection.csv $OUT/pmc_mfma_counter_collection.csv $OUT/pmc_flops_counter_collection.csv
python3 bench.py --steps 10 --warmup 2 --dtype f64 --no-cpu-baseline > $OUT/bench_f64.json 2> $OUT/bench_f64.err
python3 bench.py --steps 10 --warmup 2 --mode shared --no-cpu-baseline > $OUT/bench_shared.json 2> $OUT/bench_shared.err
timeout 900 python3 tools/other_configs.py > $OUT/other_configs.txt 2>&1
rm -rf $OUT/prof $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_lane $OUT/pmc_mfma $OUT/pmc_flops
head -c 600 $OUT/bench.json; echo; head -c 400 $OUT/bench_f64.json; echo; tail -12 $OUT/other_configs.txt; tail -3 $OUT/pmc_flops.err
