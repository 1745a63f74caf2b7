#!/bin/bash
# Profiling recipe of a round (run on the GPU box from the repo root):  bash profiles/run_profiles.sh <tag>
# Writes gpurun_out/<tag>/...; copy the summaries into profiles/ afterwards (profiles/README.md).
set -o pipefail
tag=${1:-r02}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
F32="bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-probe --no-fwd-bwd-rate"
B16="bench.py --preset config2 --steps 5 --warmup 2 --no-cpu-baseline --no-probe --no-fwd-bwd-rate"
for cfg in f32 bf16; do
  if [ $cfg = f32 ]; then args=$F32; else args=$B16; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$cfg -- python3 $args > $out/stats_$cfg.log 2>&1 && echo "stats $cfg ok"
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_mfma_$cfg -- python3 $args > $out/pmc_mfma_$cfg.log 2>&1 && echo "pmc mfma $cfg ok"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch_$cfg -- python3 $args > $out/pmc_fetch_$cfg.log 2>&1 && echo "pmc fetch $cfg ok"
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write_$cfg -- python3 $args > $out/pmc_write_$cfg.log 2>&1 && echo "pmc write $cfg ok"
  python3 profiles/summarize_pmc.py $out/pmc_mfma_$cfg $out/pmc_fetch_$cfg $out/pmc_write_$cfg > $out/pmc_summary_$cfg.csv
done
python3 profiles/make_traffic_json.py $out/pmc_fetch_f32 $out/pmc_write_f32 > $out/traffic_per_launch.json
cp $out/traffic_per_launch.json profiles/traffic_per_launch.json      # bench.py reports roofline.traffic from the file stamped with these sources
python3 bench.py > $out/bench_f32.json 2> $out/bench_f32.err && echo "bench f32 ok"
python3 bench.py --preset config2 --steps 30 --warmup 5 --no-cpu-baseline > $out/bench_bf16_b2048.json 2> $out/bench_bf16.err && echo "bench bf16 ok"
python3 bench.py --preset config5 --steps 10 --warmup 3 --no-cpu-baseline --no-probe > $out/bench_w128_bf16_b1024.json 2> $out/bench_w128.err && echo "bench w128 ok"
python3 bench.py --precision bf16 --batch 256 --steps 50 --warmup 10 --no-cpu-baseline --no-probe > $out/bench_bf16_b256.json 2>/dev/null && echo "bench bf16 b256 ok"
python3 bench.py --batch 2048 --steps 10 --warmup 3 --no-cpu-baseline --no-probe > $out/bench_f32_b2048.json 2>/dev/null && echo "bench f32 b2048 ok"
