"""HBM bytes per launch of the probed kernels, from the rocprofv3 PMC passes of bench.py.

    python profiles/make_traffic_json.py <pmc_fetch_dir> <pmc_write_dir> > profiles/traffic_per_launch.json

bytes = 2 * FETCH_SIZE + WRITE_SIZE (both in KiB; FETCH_SIZE under-reports wide streaming reads by
exactly 2x on gfx950 — MI355X_MICROARCH.md, HBM section), averaged over the launches in the pass.
bench.py reads the result for `roofline.traffic` (same workload: fp32, batch 256).
"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from source_stamp import kernel_source_sha  # noqa: E402

CONV = {  # probe name -> kernel-name prefix
    "conv_fwd_L1": "void conv5x5_mfma_kernel<32, 64, 32, false, false", "conv_fwd_L2": "void conv5x5_mfma_kernel<64, 128, 16, false, false",
    "conv_fwd_L3": "void conv5x5_mfma_kernel<128, 256, 8, false, false", "conv_fwd_L4": "void conv5x5_mfma_kernel<256, 128, 4, false, false",
    "conv_dgrad_L1": "void conv5x5_mfma_kernel<64, 32, 32, false, true", "conv_dgrad_L2": "void conv5x5_mfma_kernel<128, 64, 16, false, true",
    "conv_dgrad_L3": "void conv5x5_mfma_kernel<256, 128, 8, false, true", "conv_dgrad_L4": "void conv5x5_mfma_kernel<128, 256, 4, false, true",
    "conv_wgrad_L1": "void conv5x5_wgrad_kernel<32, 64, 32,", "conv_wgrad_L2": "void conv5x5_wgrad_kernel<64, 128, 16,",
    "conv_wgrad_L3": "void conv5x5_wgrad_kernel<128, 256, 8,", "conv_wgrad_L4": "void conv5x5_wgrad_kernel<256, 128, 4,",
    "conv_fwd_L5": "void conv_up_fwd_kernel<128, 64, 4,", "conv_fwd_L6": "void conv_up_fwd_kernel<64, 32, 8,", "conv_fwd_L7": "void conv_up_fwd_kernel<32, 32, 16,",
    "conv_dgrad_L5": "void conv_up_dgrad_kernel<128, 64, 4,", "conv_dgrad_L6": "void conv_up_dgrad_kernel<64, 32, 8,", "conv_dgrad_L7": "void conv_up_dgrad_kernel<32, 32, 16,",
    "conv_wgrad_L5": "void conv_up_wgrad_kernel<128, 64, 4>", "conv_wgrad_L6": "void conv_up_wgrad_kernel<64, 32, 8>", "conv_wgrad_L7": "void conv_up_wgrad_kernel<32, 32, 16>",
}
BN_RELU_APPLY = "void bn_bwd_kernel<0, 1>"      # launched for blocks 2, 1 in that order every step (block 0: fused into E1's weight-gradient kernel)
BN_TANH_APPLY = "void bn_bwd_kernel<1, 1>"      # block 3


def load(d, counter):
    f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            per[r["Kernel_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"]) * 1024.0))
    return {k: [v for _, v in sorted(vs)] for k, vs in per.items()}


def pick(table, prefix):
    for k, v in table.items():
        if k.startswith(prefix):
            return v
    return []


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
for name, prefix in CONV.items():
    f, w = pick(fetch, prefix), pick(write, prefix)
    if f and w:
        out[name] = round(2 * sum(f) / len(f) + sum(w) / len(w))
f, w = pick(fetch, BN_RELU_APPLY), pick(write, BN_RELU_APPLY)
for i, layer in enumerate((2, 1)):
    fi, wi = f[i::2], w[i::2]
    if fi and wi:
        out[f"bn_pool_bwd_apply_L{layer}"] = round(2 * sum(fi) / len(fi) + sum(wi) / len(wi))
f, w = pick(fetch, BN_TANH_APPLY), pick(write, BN_TANH_APPLY)
if f and w:
    out["bn_pool_bwd_apply_L3"] = round(2 * sum(f) / len(f) + sum(w) / len(w))
json.dump({"workload": "bench.py, fp32, batch 256, one MI355X", "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (see profiles/README.md)",
           "formula": "2*FETCH_SIZE + WRITE_SIZE, KiB -> bytes, mean over launches",
           "kernel_source_sha256": kernel_source_sha(), "bytes_per_launch": out}, sys.stdout, indent=1)
print()
