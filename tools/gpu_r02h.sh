set -x
mkdir -p gpurun_out/r02h
Q=learning-from-sparse-demonstrations_amd/csrc/build
timeout 600 python tools/arm_steps.py > gpurun_out/r02h/arm_steps_wide.txt 2>&1; grep -v amdgpu gpurun_out/r02h/arm_steps_wide.txt | head -40
# phase-call experiment: every variant in a process of its own
for v in default $Q/tune_robotarm_pc1.so $Q/tune_robotarm_pc2.so $Q/tune_robotarm_pc3.so; do
  for dt in f32 f64; do
    timeout 120 python tools/phase_calls_probe.py robotarm $v $dt 64 50 >> gpurun_out/r02h/phase_calls.txt 2>&1; echo "rc $? robotarm $v $dt" >> gpurun_out/r02h/phase_calls.txt
  done
done
for v in default $Q/tune_quadrotor_pc3.so; do
  timeout 120 python tools/phase_calls_probe.py quadrotor $v f32 4096 50 >> gpurun_out/r02h/phase_calls.txt 2>&1; echo "rc $? quadrotor $v f32" >> gpurun_out/r02h/phase_calls.txt
done
grep -v amdgpu gpurun_out/r02h/phase_calls.txt | grep -E "^OK|^rc|Error|error|HSA" | head -40
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r02h/bench.json 2> gpurun_out/r02h/bench.err; python -c "
import json; d=json.load(open('gpurun_out/r02h/bench.json')); print(d['value'], d['config']['kernel_ms'])"
