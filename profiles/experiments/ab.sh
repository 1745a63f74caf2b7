#!/bin/bash
# A/B of two builds of the library on the same box:  bash profiles/experiments/ab.sh <old.so> [bench args...]
# (the old build is selected through CVAE_LIB; runs alternate old/new twice)
old=$1; shift
for i in 1 2; do
  CVAE_LIB=$old python bench.py --no-cpu-baseline --no-probe "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('old', d['value'], d['ms_per_step'])"
  python bench.py --no-cpu-baseline --no-probe "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('new', d['value'], d['ms_per_step'])"
done
