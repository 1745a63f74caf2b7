#!/bin/bash
# In-kernel phase timing (s_memtime stamps) of the MS-SSIM plane kernel and of six conv instantiations, inside real training steps.
#   build (CPU box):  bash profiles/experiments/stage_timing.sh build
#   run (GPU box):    bash profiles/experiments/stage_timing.sh run > gpurun_out/stage_timing.txt
# The timing builds are separate libraries under ab/ (selected through CVAE_LIB); the shipped library carries no stamps.
root=$(cd "$(dirname "$0")/../.." && pwd); cd $root
bf16="e2_fwd:32:64:32 e4_fwd:128:256:8 e2_dgrad:64:32:32 e4_dgrad:256:128:8"
f32="e2_fwd:32:64:32 e3_dgrad:128:64:16"
if [ "$1" = build ]; then
  bash profiles/experiments/variant.sh ms_timing msssim.hip -DMS_TIMING
  for v in $bf16; do IFS=: read n k c h <<< "$v"; bash profiles/experiments/variant.sh ct_$n conv_bf16.hip -DCONV_TIMING -DCONV_TIMING_KCH=$k -DCONV_TIMING_NCH=$c -DCONV_TIMING_H=$h; done
  for v in $f32; do IFS=: read n k c h <<< "$v"; bash profiles/experiments/variant.sh cf_$n conv_mfma.hip -DCONVF_TIMING -DCONVF_TIMING_KCH=$k -DCONVF_TIMING_NCH=$c -DCONVF_TIMING_H=$h; done
  exit 0
fi
echo "== MS-SSIM plane kernel, level 0, B = 2048 (cycles per plane of waves 0 / 15 of four workgroups)"
CVAE_LIB=ab/ms_timing.so python3 profiles/experiments/time_ops.py msssim 2048 64 2>/dev/null
for v in $bf16; do IFS=: read n k c h <<< "$v"; echo "== bf16 conv $n (KCH $k, NCH $c, H $h), config2 step, B = 2048 (cycles per workgroup, summed over its K stages)"; CVAE_LIB=ab/ct_$n.so python3 profiles/experiments/conv_timing.py 2048 bf16 2>/dev/null | head -8; done
for v in $f32; do IFS=: read n k c h <<< "$v"; echo "== fp32 conv $n (KCH $k, NCH $c, H $h), headline step, B = 256"; CVAE_LIB=ab/cf_$n.so python3 profiles/experiments/conv_timing.py 256 f32 2>/dev/null | head -8; done
