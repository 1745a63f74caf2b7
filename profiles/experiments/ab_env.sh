#!/bin/bash
# A/B of an environment switch on one box, same library:  bash profiles/experiments/ab_env.sh VAR=VALUE [bench args...]   (runs alternate unset / set, twice)
kv=$1; shift
for i in 1 2; do
  python bench.py --no-cpu-baseline --no-probe "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('default', d['value'], d['ms_per_step'])"
  env $kv python bench.py --no-cpu-baseline --no-probe "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$kv', d['value'], d['ms_per_step'])"
done
