#!/bin/bash
# Same-box A/B of the streaming kernels' launch shape on the detector train step: capped grid (8192 workgroups) vs one float4 per thread.
for rnd in 1 2; do
  for cap in 8192 0; do
    DIB_ELTWISE_MAX_BLOCKS=$cap python3 scratch/train_only.py 10 2>/dev/null | tail -1 | python3 -c "import ast,sys; d=ast.literal_eval(sys.stdin.read().strip()); print('round $rnd cap=$cap: %.2f ms/step' % d['ms_per_step'])"
  done
done
