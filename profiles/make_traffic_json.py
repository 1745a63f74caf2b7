"""HBM bytes per launch of the probed kernels, from the rocprofv3 PMC passes of bench.py, per workload.

    python profiles/make_traffic_json.py f32_b256_w64=<fetch_dir>,<write_dir> bf16_b2048_w64=<f>,<w> bf16_b1024_w128=<f>,<w> \\
        > profiles/traffic_per_launch.json

bytes = 2 * FETCH_SIZE + WRITE_SIZE (both in KiB; FETCH_SIZE under-reports wide streaming reads by exactly 2x on gfx950 —
MI355X_MICROARCH.md, HBM section), averaged over the launches in the pass.  The x2 is calibrated on this code's own
access shapes by the BatchNorm apply kernels, whose byte counts are known exactly: fp32 `bn_bwd_kernel<0,1>` (16-byte
float4 accesses) 168.1 MB counted vs 167.8 MB algorithmic, bf16 `bn_bwd_bf16_kernel<0,1>` (16-byte units of 8 bf16,
4 lanes per 64-byte pixel row — the access shape of the bf16 conv / weight-gradient staging loads) 503.9 MB vs 503.3 MB.
bench.py reads the result for `roofline.traffic` / `roofline_hbm.traffic` of the same workload (key = precision_bB_wW).
"""
import collections
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from source_stamp import kernel_source_sha  # noqa: E402

LAYERS = [(3, 32, 64), (32, 64, 32), (64, 128, 16), (128, 256, 8), (256, 128, 4), (128, 64, 8), (64, 32, 16), (32, 32, 32)]


def patterns(prec, width):
    """probe name -> regex on the kernel name (layer geometry scaled to `width`)."""
    k = width // 64
    p = {}
    for l in range(1, 8):
        cin, cout, h = LAYERS[l]
        h *= k
        if prec == "f32":
            if l == 4 and h == 4:                      # D0 on 4x4 images: the padding-skipping kernel
                p[f"conv_fwd_L{l}"] = rf"conv4x4_row_kernel<{cin}, {cout}, false"
                p[f"conv_dgrad_L{l}"] = rf"conv4x4_row_kernel<{cout}, {cin}, true"
                p[f"conv_wgrad_L{l}"] = rf"conv5x5_wgrad_kernel<{cin}, {cout}, {h},"
            elif l <= 4:
                # round 4: the 64-channel-tile layers run on the persistent kernel (conv_mfma_ps.hip, <KCH, NCH, H, DGRAD, NT, EPI>)
                p[f"conv_fwd_L{l}"] = rf"conv5x5_mfma_kernel<{cin}, {cout}, {h}, false, false|conv5x5_mfma_ps_kernel<{cin}, {cout}, {h}, false"
                p[f"conv_dgrad_L{l}"] = rf"conv5x5_mfma_kernel<{cout}, {cin}, {h}, false, true|conv5x5_mfma_ps_kernel<{cout}, {cin}, {h}, true"
                p[f"conv_wgrad_L{l}"] = rf"conv5x5_wgrad_kernel<{cin}, {cout}, {h},"
            else:                                      # phase-collapsed up-convs run at the stored (low) resolution
                p[f"conv_fwd_L{l}"] = rf"conv_up_fwd_kernel<{cin}, {cout}, {h // 2},"
                p[f"conv_dgrad_L{l}"] = rf"conv_up_dgrad_kernel<{cin}, {cout}, {h // 2},"
                p[f"conv_wgrad_L{l}"] = rf"conv_up_wgrad_kernel<{cin}, {cout}, {h // 2}>"
        else:
            if l == 4 and h == 4:                      # D0 on 4x4 images: the padding-skipping kernel
                p[f"conv_fwd_L{l}"] = rf"conv4x4_row_bf16_kernel<{cin}, {cout},"
                p[f"conv_dgrad_L{l}"] = rf"conv4x4_row_bf16_kernel<{cout}, {cin},"
                p[f"conv_wgrad_L{l}"] = rf"conv5x5_wgrad_tr_kernel<{cin}, {cout}, {h},"
            elif l <= 4:
                # round 4: E2..E4 forward run on the persistent kernel (conv_bf16_ps.hip); either name matches, whichever ran
                p[f"conv_fwd_L{l}"] = rf"conv5x5_bf16(_ps)?_kernel<{cin}, {cout}, {h}, \d+, [012](, \d+, 5, 0,|>)|conv5x5_bf16_big_kernel<{cin}, {cout}, {h},"
                p[f"conv_dgrad_L{l}"] = rf"conv5x5_bf16(_ps)?_kernel<{cout}, {cin}, {h}, \d+, 2(, \d+, 5, 0,|>)|conv5x5_bf16_big_kernel<{cout}, {cin}, {h},"
                p[f"conv_wgrad_L{l}"] = rf"conv5x5_wgrad_tr_kernel<{cin}, {cout}, {h},"
            else:
                p[f"conv_fwd_L{l}"] = rf"conv5x5_bf16_kernel<{cin}, {4 * cout}, {h // 2}, \d+, 2, 1, 3, 1,"
                p[f"conv_dgrad_L{l}"] = rf"conv5x5_bf16_kernel<{4 * cout}, {cin}, {h // 2}, \d+, 2, 1, 3, 2,"
                p[f"conv_wgrad_L{l}"] = rf"conv_up_wgrad_bf16_kernel<{cin}, {cout}, {h // 2}>"
    if prec == "f32":
        p["e1_fwd"] = rf"e1_fwd_kernel<{width}>"
        p["e1_wgrad"] = rf"e1_wgrad_kernel<{width}, true>"
        p["d4_fwd"] = rf"d4_fwd_pc_f32_kernel<{width}>"
        p["d4_bwd"] = rf"d4_bwd_kernel<{width}, float>"
    else:
        p["e1_fwd"] = rf"e1_fwd_bf16_kernel<{width}, 2>"
        p["e1_wgrad"] = rf"e1_wgrad_bf16_kernel<{width}, true, true>"
        p["d4_fwd"] = rf"d4_fwd_bf16_kernel<{width}>"
        p["d4_bwd"] = rf"d4_bwd_bf16_kernel<{width}>"
    p["msssim_fwd_level0"] = rf"msssim_(stream|fwd)_kernel<{width}[,>]" if width == 128 else rf"msssim_plane_kernel<{width}, false>"
    return p


def bn_names(prec):
    base = "bn_bwd_kernel" if prec == "f32" else "bn_bwd_bf16_kernel"
    return f"void {base}<0, 1>", f"void {base}<1, 1>"       # ReLU blocks 2, 1 (in launch order; block 0 is fused into E1's wgrad) / Tanh block 3


def load(d, counter):
    f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            per[r["Kernel_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"]) * 1024.0))
    return {k: [v for _, v in sorted(vs)] for k, vs in per.items()}


def pick(table, rx):
    for k, v in table.items():
        if re.search(rx, k):
            return v
    return []


def one(prec, width, fdir, wdir):
    fetch, write = load(fdir, "FETCH_SIZE"), load(wdir, "WRITE_SIZE")
    out = {}
    for name, rx in patterns(prec, width).items():
        f, w = pick(fetch, rx), pick(write, rx)
        if f and w:
            out[name] = round(2 * sum(f) / len(f) + sum(w) / len(w))
    relu, tanh = bn_names(prec)
    f, w = pick(fetch, re.escape(relu)), pick(write, re.escape(relu))
    for i, layer in enumerate((2, 1)):
        fi, wi = f[i::2], w[i::2]
        if fi and wi:
            out[f"bn_pool_bwd_apply_L{layer}"] = round(2 * sum(fi) / len(fi) + sum(wi) / len(wi))
    f, w = pick(fetch, re.escape(tanh)), pick(write, re.escape(tanh))
    if f and w:
        out["bn_pool_bwd_apply_L3"] = round(2 * sum(f) / len(f) + sum(w) / len(w))
    return out


res = {}
for arg in sys.argv[1:]:
    key, dirs = arg.split("=")
    fdir, wdir = dirs.split(",")
    m = re.fullmatch(r"(\w+)_b(\d+)_w(\d+)", key)
    res[key] = {"workload": f"bench.py, {m.group(1)}, batch {m.group(2)}, {m.group(3)}x{m.group(3)} frames, one MI355X",
                "bytes_per_launch": one(m.group(1), int(m.group(3)), fdir, wdir)}
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (see profiles/README.md)",
           "formula": "2*FETCH_SIZE + WRITE_SIZE, KiB -> bytes, mean over launches",
           "kernel_source_sha256": kernel_source_sha(), "workloads": res}, sys.stdout, indent=1)
print()
