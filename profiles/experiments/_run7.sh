export CVAE_BIG_S16=1
timeout -k 10 600 python -m pytest tests/test_gpu_bf16.py -m gpu -x -q -k "stored_operands" > gpurun_out/t_s16.log 2>&1 || { tail -40 gpurun_out/t_s16.log; exit 1; }
tail -2 gpurun_out/t_s16.log
CVAE_LIB=ab/bs_e4d.so timeout -k 10 200 python profiles/experiments/big_steptime.py 2048 > gpurun_out/s16_steps.txt 2>&1; cat gpurun_out/s16_steps.txt
for s in 1 0; do echo "S16=$s"; CVAE_BIG_S16=$s CVAE_LIB=ab/bt_e4d.so timeout -k 10 200 python profiles/experiments/big_timing.py 2048 > gpurun_out/tmp_bt.txt 2>/dev/null; sed -n 1,3p gpurun_out/tmp_bt.txt; done
