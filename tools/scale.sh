#!/bin/bash
# 1 -> 8 GPU curves on ONE 8-GPU MI355X node (the driver runs its own SCALE pass; this is the
# exact set of command lines for a maintainer with such a node).  One process per GPU (RCCL for the
# timing barrier / MAX only: frames shard with no collective on the data path, amcpy_amd/sharding.py):
# `python3 bench.py --gpus N` starts its own N ranks; LAUNCHER=torchrun puts them under
# torch.distributed.run instead.  One JSON line per run on stdout.
#
#   bash tools/scale.sh weak     # BASELINE configs[1] per GPU (6 x 26 x 4096 x 2048 each): the driver's SCALE shape
#   bash tools/scale.sh strong   # BASELINE configs[3] (6 x 26 x 65536 x 2048, 167.5 GB) split over N GPUs;
#                                # N = 1 holds all of it in one GPU's 288 GB of HBM
#   bash tools/scale.sh c4       # BASELINE configs[4] (24 mods x 26 x 4096 x 1024): modulations sharded over N GPUs
set -e
MODE=${1:-weak}
PORT=${MASTER_PORT:-29533}
STEPS=${STEPS:-20}
WARMUP=${WARMUP:-5}
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
for N in 1 2 4 8; do
  case $MODE in
    weak)   ARGS="" ;;
    strong) ARGS="--frames $((65536 / N)) --scaling strong" ;;
    c4)     ARGS="--frame-size 1024 --mods $((24 / N)) --scaling strong" ;;
    *) echo "usage: $0 weak|strong|c4" >&2; exit 2 ;;
  esac
  if [ "$N" = 1 ]; then
    python3 bench.py --gpus 1 --steps $STEPS --warmup $WARMUP --no-cpu-baseline --no-h2d $ARGS
  elif [ "${LAUNCHER:-self}" = self ]; then
    python3 bench.py --gpus $N --steps $STEPS --warmup $WARMUP $ARGS
  else
    python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT \
      bench.py --gpus $N --steps $STEPS --warmup $WARMUP $ARGS
  fi
done
