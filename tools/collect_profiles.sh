#!/bin/bash
# Copy what `tools/gpu_session.sh final r06_final` merged into gpurun_out/ (scratch) to the tracked names under profiles/.
#   bash tools/collect_profiles.sh        (here, after the gpurun call; no GPU)
set -e
P=profiles
cpset() {  # gpurun_out dir, profiles prefix
  local d=gpurun_out/$1 pre=$2
  [ -d $d ] || { echo "missing $d"; return; }
  for f in bench.json kernel_stats.csv hbm_traffic.json issue_counters.json; do [ -f $d/$f ] && cp $d/$f $P/${pre}_$f; done
  for f in $d/pmc_*_counter_collection.csv; do [ -f $f ] && cp $f $P/${pre}_$(basename $f); done
}
cpset r06 r06
cpset r06_f64 r06_f64
cpset r06_c_robotarm r06_robotarm
cpset r06_c_rocket r06_rocket
for f in bench_default bench_f32_32768 bench_f64_32768 bench_shared_one_gpu; do [ -f gpurun_out/r06_final/$f.json ] && cp gpurun_out/r06_final/$f.json $P/r06_final_$f.json; done
[ -f gpurun_out/r06_final/rccl_one_rank.txt ] && cp gpurun_out/r06_final/rccl_one_rank.txt $P/r06_rccl_one_rank.txt
for f in pytest_gpu.txt parity_floors.jsonl smoke.txt; do [ -f gpurun_out/r06_final_tier/$f ] && cp gpurun_out/r06_final_tier/$f $P/r06_final_$f; done
[ -f gpurun_out/r06_final_tier/held_out_schedule_ab.jsonl ] && cp gpurun_out/r06_final_tier/held_out_schedule_ab.jsonl $P/r06_held_out_schedule_ab.jsonl


ls -la $P | grep -c r06_
