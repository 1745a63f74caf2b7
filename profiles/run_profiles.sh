#!/bin/bash
# Profiling recipe of a round (run on the GPU box from the repo root):  bash profiles/run_profiles.sh <tag>
# Writes gpurun_out/<tag>/...; copy the summaries into profiles/ afterwards (profiles/README.md).
# Three workloads: f32 (BASELINE configs[1], batch 256), bf16 (configs[2], batch 2048), w128 (configs[4] shard, 128x128, batch 1024);
# per workload: kernel trace + stats, then three separate --pmc passes (MFMA busy; FETCH_SIZE; WRITE_SIZE).
set -o pipefail
tag=${1:-r03}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
COMMON="--no-cpu-baseline --no-probe --no-fwd-bwd-rate --no-extra-configs"
F32="bench.py --preset config1 --steps 10 --warmup 3 $COMMON"
B16="bench.py --preset config2 --steps 5 --warmup 2 $COMMON"
W128="bench.py --preset config5 --steps 4 --warmup 2 $COMMON"
for cfg in f32 bf16 w128; do
  case $cfg in f32) args=$F32;; bf16) args=$B16;; w128) args=$W128;; esac
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$cfg -- python3 $args > $out/stats_$cfg.log 2>&1 && echo "stats $cfg ok"
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_mfma_$cfg -- python3 $args > $out/pmc_mfma_$cfg.log 2>&1 && echo "pmc mfma $cfg ok"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch_$cfg -- python3 $args > $out/pmc_fetch_$cfg.log 2>&1 && echo "pmc fetch $cfg ok"
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write_$cfg -- python3 $args > $out/pmc_write_$cfg.log 2>&1 && echo "pmc write $cfg ok"
  python3 profiles/summarize_pmc.py $out/pmc_mfma_$cfg $out/pmc_fetch_$cfg $out/pmc_write_$cfg > $out/pmc_summary_$cfg.csv
  python3 profiles/kstats.py $out/stats_$cfg > $out/kernel_stats_$cfg.txt
  cp $(find $out/stats_$cfg -name "*kernel_stats.csv" | head -1) $out/kernel_stats_$cfg.csv
done
python3 profiles/make_traffic_json.py f32_b256_w64=$out/pmc_fetch_f32,$out/pmc_write_f32 bf16_b2048_w64=$out/pmc_fetch_bf16,$out/pmc_write_bf16 \
    bf16_b1024_w128=$out/pmc_fetch_w128,$out/pmc_write_w128 > $out/traffic_per_launch.json
cp $out/traffic_per_launch.json profiles/traffic_per_launch.json      # bench.py reports roofline.traffic from the file stamped with these sources
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err && echo "bench (default: headline + config2 + config5 + drop-in) ok"
python3 bench.py --precision bf16 --batch 256 --steps 50 --warmup 10 --no-cpu-baseline --no-probe > $out/bench_bf16_b256.json 2>/dev/null && echo "bench bf16 b256 ok"
python3 bench.py --batch 2048 --steps 10 --warmup 3 --no-cpu-baseline --no-probe > $out/bench_f32_b2048.json 2>/dev/null && echo "bench f32 b2048 ok"
python3 bench.py --precision bf16x9 --steps 50 --warmup 10 --no-cpu-baseline --no-probe > $out/bench_bf16x9_b256.json 2>/dev/null && echo "bench bf16x9 ok"
python3 bench.py --precision bf16x6 --steps 50 --warmup 10 --no-cpu-baseline --no-probe > $out/bench_bf16x6_b256.json 2>/dev/null && echo "bench bf16x6 ok"
# raw traces are large: keep the summaries only
rm -rf $out/stats_* $out/pmc_mfma_* $out/pmc_fetch_* $out/pmc_write_*
