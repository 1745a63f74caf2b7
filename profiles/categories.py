"""Per-step kernel time by category from a rocprofv3 --stats kernel_stats.csv:  python profiles/categories.py <csv> <steps in the trace>"""
import csv
import sys

CATS = [
    ("conv fwd/dgrad", ("conv5x5_mfma_kernel", "conv5x5_mfma_ps_kernel", "conv4x4_row_kernel", "conv4x4_row_bf16_kernel", "conv5x5_bf16_kernel", "conv5x5_bf16_ps_kernel", "conv5x5_bf16_big_kernel", "conv_up_fwd", "conv_up_dgrad")),
    ("conv wgrad", ("conv5x5_wgrad", "conv_up_wgrad")),
    ("BatchNorm/pool", ("bn_",)),
    ("E1/D4", ("e1_", "d4_")),
    ("MS-SSIM+KLD", ("msssim",)),
    ("fc/latent", ("fc_", "decin_", "latent_gemm", "bgemm")),
    ("Adam", ("adam_kernel",)),
    ("reductions, packs, finishes", ("reduce_slabs", "rows_sum", "expand_dw", "up_finish", "splitk", "collapse_w", "pack_", "colsum", "zero_gaps", "scale3")),
]
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
tot = {c: 0.0 for c, _ in CATS}
launches, other = 0.0, []
for r in rows:
    n = r["Name"]
    if "at::" in n or "rocclr" in n:
        continue
    for c, pats in CATS:
        if any(p in n for p in pats):
            tot[c] += float(r["TotalDurationNs"]) / steps / 1e3
            launches += int(r["Calls"]) / steps
            break
    else:
        other.append(n)
print(" | ".join(f"{c} {v:.0f}" for c, v in tot.items()), f"| total {sum(tot.values()):.0f} | launches {launches:.0f}")
if other:
    print("uncategorised:", other)
