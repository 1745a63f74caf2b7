#!/bin/bash
# Round-4 stage timing of the bf16 conv kernels inside a real training step (B = 2048), one box:
#   build (repo root):  variant.sh pt  conv_bf16_ps.hip -DPS_TIMING -DPS_T_KCH=32 -DPS_T_NCH=64 -DPS_T_H=32      (persistent kernel, E2 forward)
#                       variant.sh pt3 conv_bf16_ps.hip -DPS_TIMING -DPS_T_KCH=64 -DPS_T_NCH=128 -DPS_T_H=16     (persistent kernel, E3 forward)
#                       variant.sh ct  conv_bf16.hip -DCONV_TIMING -DCONV_TIMING_KCH=32 -DCONV_TIMING_NCH=64 -DCONV_TIMING_H=32    (per-tile kernel, E2 forward)
#                       variant.sh ct3 conv_bf16.hip -DCONV_TIMING -DCONV_TIMING_KCH=64 -DCONV_TIMING_NCH=128 -DCONV_TIMING_H=16   (per-tile kernel, E3 forward)
#   run:  bash profiles/experiments/r04_timing.sh > gpurun_out/r04_stage_timing.txt
echo "== persistent kernel (conv_bf16_ps.hip), E2 forward 32->64 @32x32: cycles per item (two 128-pixel tiles x 64 channels), wave 0"
CVAE_LIB=ab/pt.so python3 profiles/experiments/ps_timing.py 2048 2>/dev/null
echo "== persistent kernel, E3 forward 64->128 @16x16"
CVAE_LIB=ab/pt3.so python3 profiles/experiments/ps_timing.py 2048 2>/dev/null
echo "== per-tile kernel (conv_bf16.hip, channel-major accumulators; CVAE_CONV_PS=0), E2 forward: cycles per workgroup (same work as one item)"
CVAE_CONV_PS=0 CVAE_LIB=ab/ct.so python3 profiles/experiments/conv_timing.py 2048 bf16 2>/dev/null
echo "== per-tile kernel, E3 forward"
CVAE_CONV_PS=0 CVAE_LIB=ab/ct3.so python3 profiles/experiments/conv_timing.py 2048 bf16 2>/dev/null
echo "== store / MFMA overlap probe (profiles/experiments/store_overlap_probe.hip)"
./profiles/experiments/store_overlap_probe.bin 2>&1
