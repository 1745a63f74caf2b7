#!/usr/bin/env python3
"""bench.py — Critic-VAE train-step images/sec on synthetic 64x64x3 frames (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B_per_gpu] [--preset config2|config4|config5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with N > 1 and no torchrun environment launches the second form itself (N fresh
child processes, started before this process touches a GPU) and relays rank 0's JSON line + exit code.

One step = forward + vae_loss + backward + (one RCCL all-reduce of the flat gradient when N>1) +
Adam, through the C-ABI (critic-vae_amd FusedTrainer), inputs resident in HBM.  Workload =
BASELINE.json configs[1]: fp32, per-GPU batch 256 (weak scaling: global batch 256*N).
Rank 0 prints ONE JSON line.  At N=1 it also reports
  roofline     — the dominant kernel of the step, timed LIVE inside the timed region by a HIP event
                 pair recorded around each of its launches on the stream it runs on
                 (cvae_probe_*), priced at its algorithmic FLOPs against the fp32 MFMA peak;
  cpu_baseline — the oracle's training step timed on the host cores (bounded sample).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

FLOP_PER_IMG = {64: 1.43762e9, 128: 5.75049e9}   # fwd + dgrad + wgrad of 9 convs + 3 linears
PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md, matrix fp32
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md, dense bf16
LAYERS = [(3, 32, 64, 0), (32, 64, 32, 0), (64, 128, 16, 0), (128, 256, 8, 0),
          (256, 128, 4, 0), (128, 64, 8, 1), (64, 32, 16, 1), (32, 32, 32, 1), (32, 3, 64, 1)]
KINDS = ("conv_fwd", "conv_dgrad", "conv_wgrad")
PROBE_IDS = [k * 9 + l for k in range(3) for l in range(1, 8)]
BN_PROBE_IDS = [27, 28, 29, 30]   # BatchNorm+pool backward apply kernel of encoder block 0..3 (HBM-bound)
BN_CH_H = [(32, 64), (64, 32), (128, 16), (256, 8)]
PEAK_HBM_GBPS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (about 6.3 TB/s achievable)
TRAFFIC_JSON = os.path.join(ROOT, "profiles", "traffic_per_launch.json")


def conv_flops(layer, B, width=64):
    """Algorithmic FLOPs of one pass (forward, dgrad or wgrad) of conv `layer` on B images."""
    cin, cout, h, _ = LAYERS[layer]
    h = h * width // 64
    return 2.0 * 25 * cin * cout * h * h * B


def probe_name(pid):
    if pid >= 27:
        return f"bn_pool_bwd_apply_L{pid - 27}"
    return f"{KINDS[pid // 9]}_L{pid % 9}"


def bn_apply_bytes(layer, B, width=64, elem_bytes=4.0):
    """Algorithmic HBM bytes of one BatchNorm+pool backward apply launch: read y and the pooled a, da;
    write dy: (2 + 2/4) * B*H*H*C elements of 4 bytes (fp32 storage) or 2 (precision mode bf16)."""
    c, h = BN_CH_H[layer]
    h = h * width // 64
    return 2.5 * B * h * h * c * elem_bytes


def measured_traffic(name):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/: separate --pmc runs,
    FETCH_SIZE x2 + WRITE_SIZE as MI355X_MICROARCH.md prescribes for gfx950).  The JSON carries the sha256
    of the kernel sources it was measured on; if the sources have changed since, the figure is stale and
    None is returned (never a number measured on other kernels)."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "profiles"))
        from source_stamp import kernel_source_sha
        with open(TRAFFIC_JSON) as f:
            j = json.load(f)
        if j.get("kernel_source_sha256") != kernel_source_sha(ROOT):
            return None
        return j["bytes_per_launch"].get(name)
    except (OSError, ValueError, KeyError, ImportError):
        return None


def cpu_baseline(B, steps=None, width=64):
    """The oracle's training step (same ATen CPU kernels as the reference, equality pinned by
    tests/golden) timed on this host: bounded sample of the same workload."""
    from critic_vae_amd import synth
    from oracle import cvae_oracle as orc
    if steps is None:                                   # about 10-30 s of CPU work whatever the batch
        steps = max(1, min(5, 1280 * 64 * 64 // (B * width * width)))
    threads = min(len(os.sched_getaffinity(0)), 16)     # the 1-GPU box's CPU share is 16 cores
    torch.set_num_threads(threads)
    p = orc.to_torch(synth.make_params(0, width), requires_grad=True)
    bn = orc.new_bn_state(p)
    st = {}
    xs = torch.rand(B, 3, width, width, generator=torch.Generator().manual_seed(1))
    pr = torch.rand(B, 1, generator=torch.Generator().manual_seed(2))
    ep = torch.randn(B, 32, generator=torch.Generator().manual_seed(3))

    def one():
        orc.zero_grad(p)
        orc.train_step(p, xs, pr, ep, bn_state=bn)
        orc.adam_step(p, st)

    one()
    t0 = time.time()
    for _ in range(steps):
        one()
    dt = time.time() - t0
    return {"value": round(B * steps / dt, 2), "unit": "images/s", "cores": torch.get_num_threads(),
            "kind": "port", "sample": f"{steps} oracle train steps (fwd+loss+bwd+Adam) at batch {B}, fp32, "
                                      f"after 1 warm-up; {dt:.1f}s of CPU work"}


PRESETS = {   # BASELINE.json configs beyond the default (configs[1] = fp32, 256/GPU, 64x64)
    "config2": dict(precision="bf16", batch=2048, width=64),     # configs[2]: 1 GPU bf16-MFMA, batch 2048
    "config4": dict(precision="bf16", batch=2048, width=64),     # configs[3]: 8 GPUs, 2048/GPU (global 16384)
    "config5": dict(precision="bf16", batch=1024, width=128),    # configs[4]: 8 GPUs, 128x128, 1024/GPU (global 8192)
}


def self_launch(n):
    """`python bench.py --gpus N` outside torchrun: start N ranks as CHILD processes (one per GPU) through
    torch.distributed.run and pass their output and exit code through.  Called before anything in this
    process initialises HIP — a process that has touched the GPU is never exec'd or re-used."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (BASELINE.json configs[1])")
    ap.add_argument("--width", type=int, choices=[64, 128], default=64, help="frame size (128: BASELINE configs[4] shape)")
    ap.add_argument("--precision", choices=["f32", "bf16", "bf16x9", "bf16x6"], default="f32",
                    help="f32 = the 1e-4-parity path (default, BASELINE configs[1]); bf16 = bf16-MFMA forward/dgrad "
                         "convs (configs[2]: use with --batch 2048)")
    ap.add_argument("--preset", choices=sorted(PRESETS), default=None,
                    help="BASELINE.json configs[2..4] workloads: sets --precision/--batch/--width")
    ap.add_argument("--allreduce-dtype", choices=["f32", "bf16"], default=None,
                    help="wire format of the gradient all-reduce at N > 1 (default f32; bf16 halves the bytes, optional per SURVEY 8e)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-probe", action="store_true")
    ap.add_argument("--no-fwd-bwd-rate", action="store_true", help="skip the extra forward+loss+backward-only loop (profiling runs)")
    args = ap.parse_args()
    if args.preset:
        for k, v in PRESETS[args.preset].items():
            setattr(args, k, v)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))

    from critic_vae_amd import dp
    from critic_vae_amd.nets import VariationalAutoencoder
    from critic_vae_amd.train import FusedTrainer

    world, rank, local = dp.init()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs MI355X GPUs (the HIP library has no CPU fallback)")
    torch.cuda.set_device(dp.device_index(local))
    dev = torch.device("cuda", dp.device_index(local))
    B = args.batch
    probing = rank == 0 and world == 1 and not args.no_probe

    def note(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    Wd = args.width
    vae = VariationalAutoencoder(width=Wd, max_batch=B, seed=0, precision=args.precision).to(dev)
    tr = FusedTrainer(vae, world_size=world, reduce_dtype=args.allreduce_dtype)
    tr.measure_exposed = world > 1
    H = vae.handle
    # synthetic inputs, resident in HBM before the timed region; each rank its own shard
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    pool = [(torch.rand(B, 3, Wd, Wd, device=dev, generator=gen), torch.rand(B, 1, device=dev, generator=gen),
             torch.randn(B, 32, device=dev, generator=gen)) for _ in range(4)]

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    note(f"model ready, batch {B}/GPU, world {world}")
    dominant, survey = None, {}
    n_survey = min(3, args.warmup) if probing else 0
    for i in range(args.warmup):
        if probing and i == args.warmup - n_survey:
            torch.cuda.synchronize()
            H.probe_config(PROBE_IDS + BN_PROBE_IDS)  # last warm-up steps: time every conv kernel in-step
        tr.step(*pool[i % len(pool)])
    torch.cuda.synchronize()
    if probing and n_survey:
        for pid in PROBE_IDS + BN_PROBE_IDS:
            ms = H.probe_read(pid)
            if ms:
                survey[pid] = sum(ms) / len(ms)
        dominant = max((p for p in survey if p < 27), key=survey.get)
        dominant_hbm = max((p for p in survey if p >= 27), key=survey.get)
        H.probe_config([dominant, dominant_hbm])      # timed region: the dominant MFMA and HBM kernels only
        note(f"dominant kernel {probe_name(dominant)} ({survey[dominant] * 1e3:.1f} us)")
    tr.exposed_us()                       # drop the warm-up samples
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        scal = tr.step(*pool[i % len(pool)])
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    dt = dp.max_over_ranks(time.perf_counter() - t0, dev)
    loss = float(scal[0].item())
    exposed = tr.exposed_us()
    dist_info = None
    if world > 1:
        # what the collectives really ran on: ranks counted by an all-reduce, backend, device of every rank
        ones = torch.ones(1, device=dev)
        torch.distributed.all_reduce(ones)
        mine = torch.zeros(world, dtype=torch.int64, device=dev)
        mine[rank] = dev.index
        torch.distributed.all_reduce(mine)
        dist_info = {"backend": torch.distributed.get_backend(), "ranks_counted_by_allreduce": int(ones.item()),
                     "world_size": torch.distributed.get_world_size(), "rank_devices": [int(v) for v in mine.tolist()],
                     "allreduce_exposed_us": None if exposed is None else round(exposed, 1),
                     "allreduce_dtype": tr.reduce_dtype,
                     "allreduce_bytes": int(tr.grads.numel() * (2 if tr.reduce_dtype == "bf16" else 4))}
    note(f"timed {args.steps} steps in {dt:.3f}s, loss {loss}")

    res = {
        "metric": f"VAE train-step images/sec on {Wd}x{Wd}x3 frames",
        "value": round(world * B * args.steps / dt, 1), "unit": "images/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None,
        "dtype": {"f32": "f32", "bf16": "bf16 (MFMA operands of every conv pass incl. E1/D4; activations and activation gradients stored as bf16; fp32 accumulate, BatchNorm statistics, loss, master weights, gradients, Adam)",
                  "bf16x9": "f32 emulated: 3-way exact bf16 operand splits, 9 bf16 MFMAs per product block (fwd+dgrad of "
                            "E2-E4, D0), f32 MFMA wgrad, f32 elsewhere",
                  "bf16x6": "f32 emulated: 3-way exact bf16 operand splits, the 6 leading partial products (fwd+dgrad of "
                            "E2-E4, D0), f32 MFMA wgrad, f32 elsewhere"}[args.precision],
        "data": "synthetic",
        "config": {"workload": ({"f32": "BASELINE.json configs[1]: fp32", "bf16": "BASELINE.json configs[2]" + ("/configs[3] per-GPU" if world > 1 else "") + ": bf16-MFMA",
                                 "bf16x9": "BASELINE.json configs[1] workload, fp32 emulated by 3-way bf16 splits",
                                 "bf16x6": "BASELINE.json configs[1] workload, fp32 emulated by 3-way bf16 splits (6 products)"}[args.precision]
                                if Wd == 64 else f"BASELINE.json configs[4] frame size (128x128), {args.precision}")
                               + " train step (fwd+MS-SSIM/KLD loss+bwd+Adam), "
                               f"batch {B}/GPU, {Wd}x{Wd}x3 frames + critic scalars", "global_batch": world * B,
                   "frame": f"{Wd}x{Wd}x3", "parallelism": f"dp{world}", "optimizer": "fused flat Adam",
                   "grad_allreduce": ("3 buckets overlapped with backward" if tr.overlap else "single, after backward") if world > 1 else "none",
                   "final_loss": loss, "loss_finite": bool(loss == loss and abs(loss) != float("inf")),
                   },
    }
    if dist_info:
        res["config"]["distributed"] = dist_info
        res["rccl_ranks"] = dist_info["ranks_counted_by_allreduce"] if dist_info["backend"] == "nccl" else 0
    if rank == 0 and world == 1:
        res["config"]["whole_step_algorithmic_TFLOPs"] = round(B * args.steps / dt * FLOP_PER_IMG[Wd] / 1e12, 2)
        if dominant is not None:
            ms = H.probe_read(dominant)
            ms_hbm = H.probe_read(dominant_hbm)
            H.probe_config([])
            sec = sum(ms) / len(ms) * 1e-3
            fl = conv_flops(dominant % 9, B, Wd)
            on_bf16 = (args.precision == "bf16" and 1 <= dominant % 9 <= 7) or (args.precision in ("bf16x9", "bf16x6") and dominant // 9 < 2 and 1 <= dominant % 9 <= 4)
            peak = PEAK_BF16_MFMA_TFLOPS if on_bf16 else PEAK_FP32_MFMA_TFLOPS
            res["roofline"] = {
                "bound": "mfma", "kernel": probe_name(dominant), "achieved": round(fl / sec / 1e12, 2),
                "peak": peak, "unit": "TFLOP/s", "frac": round(fl / sec / 1e12 / peak, 4),
                "traffic": measured_traffic(probe_name(dominant)) if B == 256 and Wd == 64 and args.precision == "f32" else None,
                "avg_launch_us": round(sec * 1e6, 2), "launches_timed": len(ms),
                "algorithmic_flops_per_launch": fl,
                "in_step_TFLOPs_all_conv_kernels": {probe_name(k): round(conv_flops(k % 9, B, Wd) / (v * 1e-3) / 1e12, 1)
                                                    for k, v in sorted(survey.items()) if k < 27}}
            if ms_hbm:          # second roofline: the largest HBM-bound kernel of the step
                sh = sum(ms_hbm) / len(ms_hbm) * 1e-3
                by = bn_apply_bytes(dominant_hbm - 27, B, Wd, 2.0 if args.precision == 'bf16' else 4.0)
                res["roofline_hbm"] = {
                    "bound": "hbm", "kernel": probe_name(dominant_hbm), "achieved": round(by / sh / 1e9, 1),
                    "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": round(by / sh / 1e9 / PEAK_HBM_GBPS, 4),
                    "traffic": measured_traffic(probe_name(dominant_hbm)) if B == 256 and Wd == 64 else None,
                    "avg_launch_us": round(sh * 1e6, 2), "launches_timed": len(ms_hbm),
                    "algorithmic_bytes_per_launch": by}
        if not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(B, width=Wd)
            if B != 32:       # the reference's own CPU-runnable case (configs[0], batch 32) beside it
                res["cpu_baseline"]["batch32"] = cpu_baseline(32, steps=10, width=Wd)
    # SURVEY 8d also asks for the C-ABI path proper (forward + loss + backward, no all-reduce / optimizer)
    fb_steps = 0 if args.no_fwd_bwd_rate else max(5, args.steps // 4)
    x0, p0, e0 = pool[0]
    theta = vae.theta.data

    def fwd_loss_bwd():
        H.forward(B, x0, p0, e0, theta, vae.bn_state, tr.mu, tr.logvar, tr.recon, tr.ws, train=True)
        H.loss(B, x0, tr.mu, tr.logvar, tr.recon, tr.ws, tr.scalars, tr.d_recon, tr.d_mu, tr.d_logvar)
        H.backward(B, x0, p0, e0, theta, tr.logvar, tr.recon, tr.d_recon, tr.d_mu, tr.d_logvar, tr.ws, tr.grads)

    if fb_steps:
        fwd_loss_bwd()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(fb_steps):
            fwd_loss_bwd()
        torch.cuda.synchronize()
        res["config"]["fwd_loss_bwd_only_images_per_s_per_gpu"] = round(B * fb_steps / (time.perf_counter() - t1), 1)

    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
