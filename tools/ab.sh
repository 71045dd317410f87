#!/bin/bash
# A/B on one box: _base/ (a git worktree of the baseline commit with its library built in place; git-ignored) against
# the working tree, alternating runs of one command, N rounds.
#   set-up: git worktree add _base <baseline commit> && make -C _base/gan_amd/csrc
#   usage:  tools/ab.sh [rounds] [command ...]      default command: python bench.py --no-cpu-baseline --no-exact-f32-time
# The command must print one JSON line with "ms_per_step".
n=${1:-3}; shift
cmd=("$@")
[ ${#cmd[@]} -eq 0 ] && cmd=(python bench.py --no-cpu-baseline --no-exact-f32-time --steps 100)
ms() { python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for i in $(seq $n); do
  a=$(cd _base && timeout -k 10 200 "${cmd[@]}" 2>/dev/null | ms)
  b=$(timeout -k 10 200 "${cmd[@]}" 2>/dev/null | ms)
  echo "round $i: baseline $a ms/step, working tree $b ms/step"
done
