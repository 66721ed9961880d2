#!/bin/bash
# same-box A/B: the shipped (tuned) MIOpen dbs vs the find-db of before the tuning (scratch/old_miopen_db)
cd $GRAFT_REPO_ROOT
for v in new old new old; do
  if [ $v = old ]; then D=/tmp/old_db_$$; rm -rf $D; mkdir -p $D; cp scratch/old_miopen_db/* $D/; export MIOPEN_USER_DB_PATH=$D; else unset MIOPEN_USER_DB_PATH; fi
  echo -n "$v dbs: "; python3 scratch/train_only.py 12 2>&1 | tail -1 | python3 -c "import sys,ast; d=ast.literal_eval(sys.stdin.read()); print('train step %.2f ms'%d['ms_per_step'])"
done
