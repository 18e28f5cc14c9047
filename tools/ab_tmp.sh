cd /root/repo
timeout -k 10 300 python -u -m pytest tests/test_gpu_heads_fusion.py -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | tail -3 && \
timeout -k 10 200 python tools/net_latency.py 64 65 96 128 129 192 256 2>&1 | grep rows && \
timeout -k 10 200 python tools/search_latency.py f32s 64 128 2>&1 | grep f32s
