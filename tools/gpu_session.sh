#!/bin/bash
# GPU sessions of a round as ONE parameterised script (the 27 one-off tools/gpu_r03_*.sh of round 3 are in the history).
# usage (on the GPU box, via gpurun):  bash tools/gpu_session.sh <session> [tag]     -> gpurun_out/<tag>/
#   quick    bench lines fp32 + fp64 (no CPU leg) and the parity tests that pin the auxiliary sweeps
#   tier     smoke() + the whole -m gpu tier with the parity-floor report
#   profile  tools/gpu_profile.sh <tag> $CFG (rocprofv3 kernel stats + PMC passes of one bench.py command)
#   configs  bench.py --config robotarm / rocket: lines with CPU legs + their evidence sets
#   f64      the same for bench.py --dtype f64
#   ab       tools/ab_variants.py run <names...>   (variants built beforehand with `ab_variants.py build`)
S=${1:-quick}; TAG=${2:-r06_$S}; OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
line() { python3 -c "
import json,sys
o=json.load(open(sys.argv[1])); r=o.get('roofline',{})
print(sys.argv[1].split('/')[-1], round(o['value']), o['ms_per_step'], o['config'].get('kernel_ms'), o['config'].get('oc_status_hist'), o['config'].get('aux_units_per_interval'), r.get('valu_frac'))" $1; }
case $S in
quick)
  python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_f32.json 2> $OUT/bench_f32.err; line $OUT/bench_f32.json
  python3 bench.py --steps 10 --warmup 2 --dtype f64 --no-cpu-baseline > $OUT/bench_f64.json 2> $OUT/bench_f64.err; line $OUT/bench_f64.json
  rm -f $OUT/parity_floors.jsonl
  LFSD_PARITY_REPORT=$PWD/$OUT/parity_floors.jsonl timeout 1500 python3 -m pytest tests -m gpu -q -x -k "${K:-every_output_grid or bench_seeds or outer_iteration_12 or reference_shaped or time_varying or fp64_aux}" > $OUT/pytest_gpu.txt 2>&1
  tail -5 $OUT/pytest_gpu.txt ;;
tier)
  python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
  rm -f $OUT/parity_floors.jsonl
  LFSD_PARITY_REPORT=$PWD/$OUT/parity_floors.jsonl timeout 2700 python3 -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1
  tail -5 $OUT/pytest_gpu.txt ;;
profile)      # [CFG="--config rocket"] gpu_session.sh profile <tag>: evidence set of one bench.py command (default: the headline)
  bash tools/gpu_profile.sh $TAG $CFG ;;
configs)      # bench lines of the other BASELINE configurations with their CPU legs, and their evidence sets
  for c in robotarm rocket; do
    STEPS=5 WARM=1 bash tools/gpu_profile.sh ${TAG}_$c --config $c > $OUT/profile_$c.log 2>&1
    line gpurun_out/${TAG}_$c/bench.json
  done ;;
f64)          # evidence set of the headline at the reference's precision
  STEPS=10 WARM=2 F64_FLOPS=1 bash tools/gpu_profile.sh $TAG --dtype f64 ;;
rocket)       # A/B of build variants on the rocket's cold OC solve (tools/model_ab.py build rocket <tag> ... beforehand): TAGS="product g10 ..."
  python3 tools/model_ab.py run rocket 100 1024 f32 ${TAGS:-product} > $OUT/rocket_ab.txt 2>&1; cat $OUT/rocket_ab.txt
  python3 tools/wide_clock.py run rocket 100 1024 f32 > $OUT/rocket_wide_clock.txt 2>&1; grep -c "wide clock" $OUT/rocket_wide_clock.txt; sort -t' ' -k6 -n -r $OUT/rocket_wide_clock.txt | head -4 ;;
steps)        # CFGS="robotarm rocket": per-outer-iteration kernel times and unit / iteration distributions (tools/config_steps.py)
  for c in ${CFGS:-robotarm}; do python3 tools/config_steps.py $c ${NSTEPS:-6} > $OUT/steps_$c.txt 2>&1; cat $OUT/steps_$c.txt; done ;;
final)        # the round's closing record: evidence sets (headline, fp64, the two configurations), default line, other batch / mode, RCCL with one rank, tier
  bash tools/gpu_session.sh profile r06 > $OUT/profile.log 2>&1
  bash tools/gpu_session.sh f64 r06_f64 > $OUT/f64.log 2>&1
  bash tools/gpu_session.sh configs r06_c
  python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; line $OUT/bench_default.json
  python3 bench.py --batch 32768 --steps 5 --warmup 1 --no-cpu-baseline --no-f64-leg > $OUT/bench_f32_32768.json 2> /dev/null; line $OUT/bench_f32_32768.json
  python3 bench.py --batch 32768 --steps 3 --warmup 1 --dtype f64 --no-cpu-baseline > $OUT/bench_f64_32768.json 2> /dev/null; line $OUT/bench_f64_32768.json
  python3 bench.py --mode shared --no-cpu-baseline > $OUT/bench_shared_one_gpu.json 2> /dev/null; line $OUT/bench_shared_one_gpu.json
  NCCL_DEBUG=VERSION python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 1 --mode shared --backend nccl --no-cpu-baseline --no-f64-leg > $OUT/rccl_one_rank.txt 2>&1; tail -1 $OUT/rccl_one_rank.txt | head -c 300; echo
  if [ -z "$NO_TIER" ]; then bash tools/gpu_session.sh tier ${TAG}_tier; fi ;;
ms)           # round 5: the multiple-shooting phase of the wide kernel, product vs variants (tools/model_ab.py build robotarm noms -DLFSD_MS=0; ... rocket msnewton -DLFSD_MS_NEWTON=1)
  vp() { python3 -c "
import sys; sys.path.insert(0,'tools'); import oc_trace
from lfsd_amd import models
print(oc_trace.variant_path(models.ZOO[sys.argv[1]]()[0].model_spec(), sys.argv[2]))" $1 $2; }
  for c in ${CFGS:-robotarm rocket}; do
    python3 bench.py --config $c --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_$c.json 2> $OUT/bench_$c.err; line $OUT/bench_$c.json
  done
  for v in ${VARIANTS:-noms nohall}; do
    python3 bench.py --config robotarm --steps 5 --warmup 1 --no-cpu-baseline --library $(vp robotarm $v) > $OUT/bench_robotarm_$v.json 2> $OUT/bench_robotarm_$v.err; line $OUT/bench_robotarm_$v.json
  done
  python3 tools/config_steps.py robotarm 6 > $OUT/steps_robotarm.txt 2>&1; grep "^step" $OUT/steps_robotarm.txt
  LFSD_TOOL_LIBRARY=$(vp robotarm noms) python3 tools/config_steps.py robotarm 6 > $OUT/steps_robotarm_noms.txt 2>&1; grep "^step" $OUT/steps_robotarm_noms.txt
  python3 tools/oc_trace.py run robotarm 50 1024 f32 ${NTRACE:-0} > $OUT/robotarm_trace.txt 2>&1; grep -c "wide" $OUT/robotarm_trace.txt
  python3 tools/wide_clock.py run robotarm 50 1024 f32 > $OUT/robotarm_wide_clock.txt 2>&1; grep -c "wide clock" $OUT/robotarm_wide_clock.txt; sort -t' ' -k6 -n -r $OUT/robotarm_wide_clock.txt | head -6; sort -t' ' -k6 -n $OUT/robotarm_wide_clock.txt | sed -n '500,503p'
  if [ -n "$K" ]; then
    LFSD_PARITY_REPORT=$PWD/$OUT/parity_floors.jsonl timeout 2400 python3 -m pytest tests -m gpu -q -k "$K" > $OUT/pytest_gpu.txt 2>&1
    tail -15 $OUT/pytest_gpu.txt
  fi ;;
ab)
  shift; shift; python3 tools/ab_variants.py run "$@" > $OUT/ab.txt 2>&1; cat $OUT/ab.txt ;;
*) echo "unknown session $S"; exit 2 ;;
esac
