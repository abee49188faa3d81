#!/bin/bash
# Rehearsal of bench.py's multi-rank path on a one-GPU box (all ranks on cuda:0, gloo): the assembled frame must
# equal the single-rank frame byte for byte.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/mgpu
timeout -k 10 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --save-png gpurun_out/mgpu/n1.png | cut -c1-200 || exit 1
for n in 2 3; do
RT_BENCH_ONE_DEVICE=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2951$n bench.py --gpus $n --steps 1 --warmup 1 --save-png gpurun_out/mgpu/n$n.png 2>gpurun_out/mgpu/n$n.err | cut -c1-400 || exit 1
cmp gpurun_out/mgpu/n1.png gpurun_out/mgpu/n$n.png && echo "N=$n frame identical to N=1"
done
