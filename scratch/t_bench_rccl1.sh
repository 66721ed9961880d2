#!/bin/bash
# bench.py's RCCL branch on a 1-GPU box: one rank, real process group (init, all_gather, barrier, all_reduce MAX, DDP wrapper,
# distributed engine mode).  gpurun -- bash scratch/t_bench_rccl1.sh
export DIB_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 LOCAL_WORLD_SIZE=1
python3 bench.py --gpus 1 --repeats 10 --no-cpu-baseline --no-eval-sweep > gpurun_out/bench_rccl1.json 2> gpurun_out/bench_rccl1.err
echo rc=$?
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_rccl1.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "n_gpus", "world_size_seen_by_rccl", "ms_per_step")})
print("train_step", d["train_step"]["value"], d["train_step"]["parallelism"], "train_e2e", d["train_e2e"]["value"])
PY
