#!/bin/bash
# un-profiled step rate by which layers run on the persistent conv kernel:  bash profiles/experiments/ps_mask_sweep.sh [bench args]
for rep in 1 2; do
for m in 0 1 2 4 8 16 32 7 63; do
  CVAE_CONV_PS=$m python bench.py --no-cpu-baseline --no-probe --no-fwd-bwd-rate --no-extra-configs "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('mask $m', d['value'], d['ms_per_step'])"
done; done
