#!/bin/bash
# Two ranks on the ONE GPU a gpurun box has: proves the nccl (= RCCL) path of bench.py initialises, sees world_size 2 and moves the
# flat gradient buffers — or records RCCL's refusal to put two ranks on one device.  Not a performance number.
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --train-step --steps 2 --warmup 1 --batch 2 --train-nrr 64
