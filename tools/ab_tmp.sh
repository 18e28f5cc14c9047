cd /root/repo
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 600 --timeout-method thread 2>&1 | tail -4 && \
timeout -k 10 200 python tools/search_latency.py > gpurun_out/search_latency.txt 2>&1 && \
timeout -k 10 200 python tools/net_latency.py 1 8 32 64 65 128 192 256 384 512 768 1024 1536 2048 > gpurun_out/net_latency.txt 2>&1 && \
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-extra-lines --no-cpu-baseline 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read()); print('bench', round(d['value']), d['ms_per_step'], d['kernel_ms'])"
