#!/bin/bash
# A/B on one box: the round-1 tree (_base/, a git worktree of the baseline commit, library built in place) against
# the working tree, bench.py alternating, N rounds.  usage: tools/ab.sh [rounds]
# set-up: git worktree add _base <baseline commit> && make -C _base/gan_amd/csrc   (_base/ is git-ignored)
n=${1:-3}
for i in $(seq $n); do
  a=$(cd _base && timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  b=$(timeout -k 10 200 python bench.py --no-cpu-baseline --no-exact-f32-time 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "round $i: baseline $a ms/step, working tree $b ms/step"
done
