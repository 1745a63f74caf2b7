export CVAE_BIG_S16=1
timeout -k 10 600 python -m pytest tests/test_gpu_bf16.py -m gpu -x -q -k "stored_operands or bitwise or full_size" > gpurun_out/t_s16.log 2>&1 || { tail -40 gpurun_out/t_s16.log; exit 1; }
tail -3 gpurun_out/t_s16.log
bash profiles/experiments/kprof.sh s16/on - "conv5x5_bf16_big" --preset config2 --no-extra-configs
unset CVAE_BIG_S16
bash profiles/experiments/kprof.sh s16/off - "conv5x5_bf16_big" --preset config2 --no-extra-configs
