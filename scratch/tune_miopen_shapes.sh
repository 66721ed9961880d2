#!/bin/bash
# MIOpen solver tuning (MIOPEN_FIND_ENFORCE=SEARCH) of the train step's convolutions at the most common padded batch shapes of COCO
# training, on top of the shipped dbs; before / after per shape.  Results: gpurun_out/miopen_tune_shapes/ (copy the *.udb.txt / *.ufdb.txt
# into detectinblur_amd/miopen_db/).
#   gpurun --timeout 3300 -- bash scratch/tune_miopen_shapes.sh "800 1216" "800 1088" ...
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
D=$GRAFT_REPO_ROOT/gpurun_out/miopen_tune_shapes; rm -rf $D; mkdir -p $D; cp detectinblur_amd/miopen_db/* $D/
for shape in "$@"; do
  echo "== $shape: before"; python3 scratch/t_shape_steps.py $shape 10 2>/dev/null | tail -1
  t0=$(date +%s)
  MIOPEN_USER_DB_PATH=$D MIOPEN_FIND_ENFORCE=3 timeout ${TUNE_SECONDS:-900} python3 scratch/t_shape_steps.py $shape 1 > gpurun_out/miopen_tune_shapes.log 2>&1; echo "tuning run: rc $? after $(( $(date +%s) - t0 )) s"
  echo "== $shape: with the tuned dbs"; MIOPEN_USER_DB_PATH=$D python3 scratch/t_shape_steps.py $shape 10 2>/dev/null | tail -1
done
wc -l $D/*
