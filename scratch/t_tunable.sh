#!/bin/bash
# PyTorch TunableOp (hipBLASLt / rocBLAS solution search per GEMM shape) on the batch-1 inference trunk: baseline, tuning run, tuned run.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
echo "== baseline"; WARM=2 N=30 python3 scratch/t_graph_speed.py 2>&1 | grep -E "trunk graph replay|graphed forward:"
echo "== tuning"; PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_FILENAME=gpurun_out/tunableop_b1.csv PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=10 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=2 \
  WARM=2 N=30 timeout 1500 python3 scratch/t_graph_speed.py 2>&1 | grep -E "trunk graph replay|graphed forward:|rror" | head
ls -la gpurun_out/tunableop_b1*.csv; wc -l gpurun_out/tunableop_b1*.csv
echo "== tuned, no further tuning"; PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=0 PYTORCH_TUNABLEOP_FILENAME=gpurun_out/tunableop_b1.csv \
  WARM=2 N=30 python3 scratch/t_graph_speed.py 2>&1 | grep -E "trunk graph replay|graphed forward:|rror" | head
