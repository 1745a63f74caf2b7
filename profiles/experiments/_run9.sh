export CVAE_BIG_S16=1
timeout -k 10 800 python -m pytest tests/test_gpu_bf16.py -m gpu -x -q > gpurun_out/t_s16.log 2>&1 || { tail -40 gpurun_out/t_s16.log; exit 1; }
tail -2 gpurun_out/t_s16.log
unset CVAE_BIG_S16
one() { echo -n "S16=$1: "; CVAE_BIG_S16=$1 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-probe --no-fwd-bwd-rate --no-extra-configs --preset config2 2>/dev/null > gpurun_out/tmp_b.json; python3 -c "import sys,json; d=json.loads(open('gpurun_out/tmp_b.json').read()); print(d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do one 0; one 1; done
CVAE_BIG_S16=1 bash profiles/experiments/kprof.sh s16/on - "conv5x5_bf16_big" --preset config2 --no-extra-configs
CVAE_BIG_S16=0 bash profiles/experiments/kprof.sh s16/off - "conv5x5_bf16_big" --preset config2 --no-extra-configs
