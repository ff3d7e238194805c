#!/bin/bash
# Can two RCCL ranks share ONE GPU?  (bench.py --gpus 2 with both ranks pinned to device 0 and the real nccl backend)
O=gpurun_out/r04_nccl2; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
VSSR_LOCAL_DEVICE=0 NCCL_DEBUG=WARN timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
   bench.py --gpus 2 --steps 3 --warmup 1 --chains-per-gpu 16 --no-cpu-baseline > $O/out.txt 2> $O/err.txt
echo "rc=$?" >> $O/out.txt
tail -5 $O/out.txt; grep -i "duplicate\|error\|invalid" $O/err.txt | head -10
