#!/bin/bash
# TunableOp refill with cold operands (rotating buffer > Infinity Cache) and longer timing; same-box A/B against the shipped file
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
echo -n "shipped csv: "; python3 scratch/train_only.py 12 2>&1 | tail -1 | python3 -c "import sys,ast; d=ast.literal_eval(sys.stdin.read()); print('train step %.2f ms'%d['ms_per_step'])"
python3 scratch/t_eval_anatomy.py 2>/dev/null | head -1
sed -i 's/PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=15 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=3/PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=40 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=5 PYTORCH_TUNABLEOP_ROTATING_BUFFER_SIZE=512/' scratch/fill_tunableop.sh
bash scratch/fill_tunableop.sh 2>&1 | tail -2
cp gpurun_out/tunableop_results.csv detectinblur_amd/tunableop/tunableop_results.csv
echo -n "refilled csv: "; python3 scratch/train_only.py 12 2>&1 | tail -1 | python3 -c "import sys,ast; d=ast.literal_eval(sys.stdin.read()); print('train step %.2f ms'%d['ms_per_step'])"
python3 scratch/t_eval_anatomy.py 2>/dev/null | head -1
