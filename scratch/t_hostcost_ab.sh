#!/bin/bash
# Host cost of one eager blur step, the tree under scratch/old_tree (a `git archive` of the previous commit) against this one.
for rnd in 1 2; do
  (cd scratch/old_tree && python3 ../t_hostcost.py 2>/dev/null | grep -E "^(800|70)" | sed "s/^/old  /")
  python3 scratch/t_hostcost.py 2>/dev/null | grep -E "^(800|70)" | sed "s/^/new  /"
done
