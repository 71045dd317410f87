#!/bin/bash
# A/B of two builds of the library on one box: GAN_HIP_LIB=<baseline .so> against the in-tree build, alternating runs of
# one command, N rounds.
#   usage: tools/lib_ab.sh <baseline .so> [rounds] [command ...]   default command: bench.py without the CPU legs
# The command must print one JSON line with "ms_per_step".
base=$(readlink -f "$1"); shift
n=${1:-3}; shift
cmd=("$@")
[ ${#cmd[@]} -eq 0 ] && cmd=(python bench.py --no-cpu-baseline --no-exact-f32-time --steps 100)
ms() { python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for i in $(seq $n); do
  a=$(GAN_HIP_LIB=$base timeout -k 10 200 "${cmd[@]}" 2>/dev/null | ms) || exit 1
  b=$(timeout -k 10 200 "${cmd[@]}" 2>/dev/null | ms) || exit 1
  echo "round $i: baseline $a ms/step, working tree $b ms/step"
done
