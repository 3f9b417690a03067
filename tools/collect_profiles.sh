#!/bin/bash
# Copy what `tools/gpu_session.sh final r05_final` merged into gpurun_out/ (scratch) to the tracked names under profiles/.
#   bash tools/collect_profiles.sh        (here, after the gpurun call; no GPU)
set -e
P=profiles
cpset() {  # gpurun_out dir, profiles prefix
  local d=gpurun_out/$1 pre=$2
  [ -d $d ] || { echo "missing $d"; return; }
  for f in bench.json kernel_stats.csv hbm_traffic.json issue_counters.json; do [ -f $d/$f ] && cp $d/$f $P/${pre}_$f; done
  for f in $d/pmc_*_counter_collection.csv; do [ -f $f ] && cp $f $P/${pre}_$(basename $f); done
}
cpset r05 r05
cpset r05_f64 r05_f64
cpset r05_c_robotarm r05_robotarm
cpset r05_c_rocket r05_rocket
for f in bench_default bench_f32_32768 bench_f64_32768 bench_shared_one_gpu; do [ -f gpurun_out/r05_final/$f.json ] && cp gpurun_out/r05_final/$f.json $P/r05_final_$f.json; done
[ -f gpurun_out/r05_final/rccl_one_rank.txt ] && cp gpurun_out/r05_final/rccl_one_rank.txt $P/r05_rccl_one_rank.txt
for f in pytest_gpu.txt parity_floors.jsonl smoke.txt; do [ -f gpurun_out/r05_final_tier/$f ] && cp gpurun_out/r05_final_tier/$f $P/r05_final_$f; done
[ -f gpurun_out/r05_final_tier/held_out_schedule_ab.jsonl ] && cp gpurun_out/r05_final_tier/held_out_schedule_ab.jsonl $P/r05_held_out_schedule_ab.jsonl


ls -la $P | grep -c r05_
