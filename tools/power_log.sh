#!/bin/bash
# Samples the GPU's power draw, clocks and temperature (rocm-smi, read-only) while the headline benchmark runs.
# Usage (GPU box, repository root): bash tools/power_log.sh > gpurun_out/power.txt
cd "$(dirname "$0")/.."
rocm-smi -M 2>&1 | grep -i "max\|cap" | head -4
echo "--- idle"
rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -i "power\|sclk\|mclk\|fclk\|junction\|Temperature (Sensor memory)" | head -8
timeout -k 10 200 python bench.py --steps 12 --warmup 2 --no-extra-lines --no-cpu-baseline > /tmp/power_bench.json 2>/dev/null &
pid=$!
sleep 20          # import torch, engine set-up, warm-up
echo "--- under load, the power query's whole answer"
rocm-smi -P 2>&1 | grep -v "^$" | head -12
(command -v amd-smi > /dev/null && amd-smi metric --power --clock 2>&1 | head -40) || true
for i in 1 2 3 4 5 6 7 8; do
  echo "--- under load, sample $i"
  rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -i "power\|sclk\|mclk\|fclk\|junction\|Temperature (Sensor memory)" | head -8
  sleep 1
done
wait $pid
python3 -c "
import json; d = json.load(open('/tmp/power_bench.json')); print('bench:', round(d['value']), 'games/s', d['kernel_ms'], d['roofline']['shader_clock_mhz'])"
