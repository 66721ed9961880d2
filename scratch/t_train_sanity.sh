#!/bin/bash
# 30 iterations of the real driver at 800 x 1333, b = 8, with the shipped (tuned) dbs and with the find-db of before the tuning:
# the loss trajectories must look alike (same seeds; run-to-run variation of the weight-gradient kernels aside)
cd $GRAFT_REPO_ROOT
for v in new old; do
  if [ $v = old ]; then D=/tmp/old_db_$$; rm -rf $D; mkdir -p $D; cp scratch/old_miopen_db/* $D/; export MIOPEN_USER_DB_PATH=$D; export DIB_NO_TUNABLEOP=1; else unset MIOPEN_USER_DB_PATH DIB_NO_TUNABLEOP; fi
  echo "== $v"
  python3 -m detectinblur_amd.train --synthetic --synthetic_images 256 --synthetic_size 800 1333 -b 8 -j 4 --epochs 1 --blur_train --gpu_blur --param_index 1 --low_exposure \
     --expand_target_boxes --early_stop 30 --lr 0.01 --print_freq 10 --output_dir /tmp/w_$v 2>&1 | grep -E "^Epoch: \[0\]" | cut -c1-220
done
