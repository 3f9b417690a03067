#!/bin/bash
# Evidence for profiles/: bench line, rocprofv3 kernel stats of `bench.py`, then the PMC passes (separate runs, --kernel-trace
# only beside --pmc), folded per FULL-BATCH dispatch.
# usage (on the GPU box, via gpurun):  [LFSD_COMMIT=<short hash>] bash tools/gpu_profile.sh <tag> [bench.py arguments, e.g. --config rocket]
#   -> gpurun_out/<tag>/{bench.json, kernel_stats.csv, pmc_*_counter_collection.csv, hbm_traffic.json, issue_counters.json}
TAG=${1:-r06}; shift
ARGS="$@"
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
STEPS=${STEPS:-20}; WARM=${WARM:-5}
MODEL=quadrotor; case "$ARGS" in *robotarm*) MODEL=robotarm;; *rocket*) MODEL=rocket;; esac
case "$ARGS" in *"--dtype f64"*) export LFSD_PROFILE_DTYPE=f64;; esac      # the folds keep the fp32 seeding solve apart from the fp64 kernel
export LFSD_KERNEL_RESOURCES=${LFSD_KERNEL_RESOURCES:-profiles/r06_kernel_resources_$MODEL.json}      # (tools/kernel_resources.py <model> --json, made at build time)
python3 bench.py --steps $STEPS --warmup $WARM $ARGS > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT/prof -o $TAG --output-format csv -- python3 bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --no-f64-leg $ARGS > $OUT/bench_prof.json 2> $OUT/prof.err
pass() {  # name, counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d $OUT/pmc_$name -o $name --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f64-leg $ARGS > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err
  cp $(find $OUT/pmc_$name -name "*counter_collection.csv") $OUT/pmc_${name}_counter_collection.csv
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass lane SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS
pass mfma SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
pass flops SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FLOPS_FP32 SQ_THREAD_CYCLES_VALU
if [ -n "$F64_FLOPS" ]; then      # fp64 workloads: the fp64 instruction classes instead
  pass flops64 SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FLOPS_FP64 SQ_THREAD_CYCLES_VALU
fi
cp $(find $OUT/prof -name "*kernel_stats.csv") $OUT/kernel_stats.csv
python3 tools/hbm_traffic.py $OUT/pmc_fetch_counter_collection.csv $OUT/pmc_write_counter_collection.csv $OUT/hbm_traffic.json
python3 tools/issue_counters.py $OUT/issue_counters.json $OUT/pmc_lane_counter_collection.csv $OUT/pmc_mfma_counter_collection.csv $OUT/pmc_flops_counter_collection.csv $(ls $OUT/pmc_flops64_counter_collection.csv 2>/dev/null)
rm -rf $OUT/prof $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_lane $OUT/pmc_mfma $OUT/pmc_flops $OUT/pmc_flops64
head -c 700 $OUT/bench.json; echo; tail -3 $OUT/pmc_flops.err
