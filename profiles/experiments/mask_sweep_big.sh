#!/bin/bash
# un-profiled step rate for several CVAE_BF16_BIG masks, alternating, one box:  bash profiles/experiments/mask_sweep_big.sh "36 60 52 44" 3 [bench args]
masks=$1; reps=$2; shift 2
for i in $(seq $reps); do
  for m in $masks; do
    echo -n "mask $m: "; CVAE_BF16_BIG=$m python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-probe --no-fwd-bwd-rate --no-extra-configs "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done
done
