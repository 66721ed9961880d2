#!/bin/bash
# same-box A/B of the shipped TunableOp choices: train step and the evaluation loop, with and without
cd $GRAFT_REPO_ROOT
for v in 0 1; do
  if [ $v = 1 ]; then export DIB_NO_TUNABLEOP=1; else unset DIB_NO_TUNABLEOP; fi
  echo "== DIB_NO_TUNABLEOP=$v"
  python3 scratch/train_only.py 12 2>&1 | tail -1 | python3 -c "import sys,ast; d=ast.literal_eval(sys.stdin.read()); print('train step %.2f ms'%d['ms_per_step'])"
  python3 scratch/t_eval_anatomy.py 2>/dev/null | head -1
done
