#!/usr/bin/env bash
# usage: bash run.sh example/clip_fdt/train_solver.py --config example/clip_fdt/config_cc3m.yaml --output_path out --batch_size 256
# one process per GPU over RCCL/xGMI (same env contract as the reference's run.sh: RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*)
GPU_NUM=${GPU_NUM:-8}
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -m torch.distributed.run --nnodes=1 --nproc-per-node "$GPU_NUM" --master-addr 127.0.0.1 \
    --master-port "${MASTER_PORT:-29500}" "$@" || exit 1
