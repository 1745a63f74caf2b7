#!/bin/bash
# In-step times of every probed kernel for several CVAE_BF16_BIG masks, one box:  bash profiles/experiments/instep_all.sh "36 60"
masks=$1; shift
for m in $masks; do
  CVAE_BF16_BIG=$m CVAE_BENCH_DETAIL=/tmp/bd_$m.json python bench.py --preset config2 --steps 20 --warmup 5 --no-cpu-baseline --no-fwd-bwd-rate --no-extra-configs "$@" > /tmp/line_$m.json 2>/dev/null
done
python3 - $masks <<'PY'
import json, sys
ms = sys.argv[1:]
tab = {}
for m in ms:
    d = json.load(open(f"/tmp/bd_{m}.json"))["detail"]["headline"]
    line = json.load(open(f"/tmp/line_{m}.json"))
    fl = 2 * 25 * 64 * 32 * 32 * 32 * 2048
    t = {}
    for k, v in d["roofline"]["in_step_TFLOPs_all_conv_kernels"].items():
        t[k] = v      # TFLOP/s (layer-dependent flops): compare ratios
    for k, v in d["roofline_hbm"]["in_step_us_GBps_all_hbm_side_kernels"].items():
        t[k] = v[0] if isinstance(v, list) else v
    t["__step_ms"] = line["ms_per_step"]
    tab[m] = t
keys = list(tab[ms[0]].keys())
print("kernel".ljust(28), *[f"mask {m}".rjust(12) for m in ms])
for k in keys:
    print(k.ljust(28), *[str(tab[m].get(k)).rjust(12) for m in ms])
PY
