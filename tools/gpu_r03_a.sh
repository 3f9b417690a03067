#!/bin/bash
# round 3, GPU session A: issue-rate probe, A/B of build variants, batch sweep, per-dispatch traffic + lane-occupancy counters
OUT=gpurun_out/r03a
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
./tools/probes/valu_issue_probe > $OUT/valu_issue_probe.txt 2>&1
cat $OUT/valu_issue_probe.txt
python3 tools/ab_variants.py run base noprefetch slp --steps 20 --batch 4096 > $OUT/ab_variants.txt 2>&1
python3 tools/ab_variants.py run base --steps 20 --batch 1024 --batch 2048 --batch 8192 >> $OUT/ab_variants.txt 2>&1
cat $OUT/ab_variants.txt
B="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o fetch --output-format csv -- $B > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o write --output-format csv -- $B > $OUT/pmc_write.json 2> $OUT/pmc_write.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS -d $OUT/pmc_lane -o lane --output-format csv -- $B > $OUT/pmc_lane.json 2> $OUT/pmc_lane.err
tail -2 $OUT/pmc_lane.err
python3 tools/hbm_traffic.py $(find $OUT/pmc_fetch -name "*counter_collection.csv") $(find $OUT/pmc_write -name "*counter_collection.csv") $OUT/hbm_traffic.json
cp $(find $OUT/pmc_lane -name "*counter_collection.csv") $OUT/pmc_lane_counter_collection.csv
cp $(find $OUT/pmc_fetch -name "*counter_collection.csv") $OUT/pmc_fetch_counter_collection.csv
cp $(find $OUT/pmc_write -name "*counter_collection.csv") $OUT/pmc_write_counter_collection.csv
python3 bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json
