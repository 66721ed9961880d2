#!/bin/bash
# Extends the SHIPPED MIOpen user find-db (detectinblur_amd/miopen_db) by every convolution shape the bench, the drivers and the
# GPU tests meet: runs them with DIB_MIOPEN_DB_INPLACE=1 (the package then works on the shipped directory instead of a private
# copy) and copies the result to gpurun_out/miopen_db_full/.   gpurun -- bash scratch/fill_miopen_db.sh ; then commit the file.
cd $GRAFT_REPO_ROOT
export DIB_MIOPEN_DB_INPLACE=1
wc -l detectinblur_amd/miopen_db/*.ufdb.txt
python3 bench.py --steps 5 --warmup 5 --repeats 3 --train-steps 3 --train-warmup 3 --e2e-steps 3 --e2e-warmup 2 --sweep-images 4 --no-cpu-baseline > /dev/null 2>&1
wc -l detectinblur_amd/miopen_db/*.ufdb.txt
python3 -m pytest tests/test_full_size_gpu.py tests/test_train_step_gpu.py tests/test_blur_estimator.py tests/test_cli_and_data.py tests/test_epilogue_gpu.py -q -m gpu > /dev/null 2>&1
wc -l detectinblur_amd/miopen_db/*.ufdb.txt
python3 -m detectinblur_amd.evaluate --synthetic --synthetic_images 6 --synthetic_size 480 640 -j 2 --blur_eval --gpu_blur --expand_target_boxes --early_stop 4 > /dev/null 2>&1
python3 -m detectinblur_amd.evaluate --synthetic --synthetic_images 6 --synthetic_size 640 480 -j 2 --blur_eval --gpu_blur --expand_target_boxes --early_stop 4 > /dev/null 2>&1
python3 -m detectinblur_amd.evaluate --synthetic --synthetic_images 6 --synthetic_size 427 640 -j 2 --blur_eval --gpu_blur --expand_target_boxes --early_stop 4 > /dev/null 2>&1
python3 -m detectinblur_amd.train --synthetic --synthetic_images 16 --synthetic_size 480 640 -b 8 -j 2 --epochs 1 --blur_train --gpu_blur --param_index 1 --low_exposure --expand_target_boxes --early_stop 2 --output_dir /tmp/w > /dev/null 2>&1
wc -l detectinblur_amd/miopen_db/*.ufdb.txt
mkdir -p gpurun_out/miopen_db_full
cp detectinblur_amd/miopen_db/*.ufdb.txt gpurun_out/miopen_db_full/
