#!/bin/bash
# MIOpen solver tuning (MIOPEN_FIND_ENFORCE=SEARCH) of the train step's convolutions: does a tuned perf-db beat the find-db's choices?
#   gpurun --timeout 3300 -- bash scratch/tune_miopen_train.sh
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
D=$GRAFT_REPO_ROOT/gpurun_out/miopen_tune; rm -rf $D; mkdir -p $D; cp detectinblur_amd/miopen_db/* $D/
echo "== baseline (shipped find-db, private copy)"; python3 scratch/train_only.py 12 2>&1 | tail -1 | python3 -c "import sys,ast; d=ast.literal_eval(sys.stdin.read()); print('train step %.2f ms'%d['ms_per_step'])"
echo "== tuning run"; t0=$(date +%s)
MIOPEN_USER_DB_PATH=$D MIOPEN_FIND_ENFORCE=3 timeout ${TUNE_SECONDS:-2400} python3 scratch/train_only.py 1 > gpurun_out/miopen_tune.log 2>&1; echo "rc $? after $(( $(date +%s) - t0 )) s"
ls -la $D; wc -l $D/*
echo "== with the tuned dbs"; MIOPEN_USER_DB_PATH=$D python3 scratch/train_only.py 12 2>&1 | tail -1 | python3 -c "import sys,ast; d=ast.literal_eval(sys.stdin.read()); print('train step %.2f ms'%d['ms_per_step'])"
