set -x
mkdir -p gpurun_out/r02b
timeout 600 python bench.py > gpurun_out/r02b/bench.json 2> gpurun_out/r02b/bench.err; echo "bench rc $?"
timeout 600 python tools/arm_steps.py > gpurun_out/r02b/arm_steps.txt 2>&1
timeout 300 python tools/diag_lockstep.py > gpurun_out/r02b/lockstep.txt 2>&1
timeout 900 python tools/other_configs.py > gpurun_out/r02b/other.txt 2>&1
nproc > gpurun_out/r02b/nproc.txt
cat gpurun_out/r02b/bench.json
