#!/bin/bash
# In-step (HIP events inside the un-profiled training step) times of the conv kernels for several CVAE_BF16_BIG masks, one box:
#   bash profiles/experiments/instep_conv.sh "0 36 60" [bench args]
masks=$1; shift
for m in $masks; do
  CVAE_BF16_BIG=$m CVAE_BENCH_DETAIL=/tmp/bd_$m.json python bench.py --preset config2 --steps 20 --warmup 5 --no-cpu-baseline --no-fwd-bwd-rate --no-extra-configs "$@" > /tmp/line_$m.json 2>/dev/null
  python3 - $m <<'PY'
import json, sys
m = sys.argv[1]
line = json.load(open(f"/tmp/line_{m}.json"))
d = json.load(open(f"/tmp/bd_{m}.json"))["detail"]["headline"]
t = d["roofline"]["in_step_TFLOPs_all_conv_kernels"]
B = 2048
fl = 2 * 25 * 64 * 32 * 32 * 32 * B      # every big conv pass of E2..E4 has the same FLOPs
us = {k: round(fl / (v * 1e12) * 1e6, 1) for k, v in t.items() if k.split("_L")[-1] in ("1", "2", "3")}
print(f"mask {m}: {line['value']} img/s {line['ms_per_step']} ms | in-step us:", us)
PY
done
