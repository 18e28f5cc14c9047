cd /root/repo
mkdir -p gpurun_out/final
timeout -k 10 400 python bench.py --gpus 1 --steps 10 --warmup 3 > gpurun_out/final/bench_line.json 2> gpurun_out/final/bench_line.err || exit 1
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 1 --steps 3 --warmup 2 --no-extra-lines --no-cpu-baseline > gpurun_out/final/bench_one_rank_rccl.json 2> gpurun_out/final/bench_one_rank_rccl.err || exit 1
NNB_BENCH_REHEARSAL_ONE_GPU=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/final/bench_two_ranks_one_gpu.json 2> gpurun_out/final/bench_two_ranks_one_gpu.err || exit 1
echo done
