for s in 1 0; do
  echo "== S16=$s"
  CVAE_BIG_S16=$s CVAE_LIB=ab/bt_e4d.so timeout -k 10 200 python profiles/experiments/big_timing.py 2048 > gpurun_out/tmp_bt.txt 2>/dev/null || exit 1
  sed -n 1,4p gpurun_out/tmp_bt.txt
done
