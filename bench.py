#!/usr/bin/env python3
"""bench.py — Critic-VAE train-step images/sec on synthetic 64x64x3 frames (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B_per_gpu] [--preset config2|config4|config5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with N > 1 and no torchrun environment launches the second form itself (N fresh
child processes, started before this process touches a GPU) and relays rank 0's JSON line + exit code.

One step = forward + vae_loss + backward + (RCCL all-reduce of the flat gradient when N>1) + Adam, through
the C-ABI (critic-vae_amd FusedTrainer), inputs resident in HBM.  Rank 0 prints ONE JSON line:

  headline (value, ms_per_step, roofline, roofline_hbm, cpu_baseline) = BASELINE.json configs[1]: fp32,
      per-GPU batch 256 (weak scaling: global batch 256*N) — unless --preset/--batch/--precision/--width ask
      for another workload;
  Every sub-workload below lands in `config` twice: as FLAT scalar keys (config2_images_per_s, config2_ms_per_step,
      config2_steps, config2_roofline_kernel / _frac / _avg_launch_us / _traffic_over_algorithmic, config2_roofline_hbm_frac,
      ... the same for config5 / bf16x9 / bf16x6, dropin_images_per_s — a record that keeps scalars only keeps these) and as
      a compact nested object; per-kernel tables and long descriptions go to config.detail_file (+ stderr).
  config.config2 / config.config5 (N = 1) = the SAME timed loop (same barrier/sync bracket, same in-step kernel
      probe, --steps/--warmup as given) run right after it in this process on BASELINE.json configs[2]
      (bf16-MFMA, batch 2048) and on the per-GPU shard of configs[4] (128x128, bf16, batch 1024), each with
      both rooflines; `side_stream_wgrad` inside each = the rate with cvae_config.overlap_wgrad = 1 (weight-gradient
      kernels on the library's side stream, bit-identical results; no rooflines: kernels share the chip);
  config.config4 / config.config5 (N > 1) = configs[3] / configs[4] per GPU, each timed with the gradient
      all-reduce in three buckets overlapped with backward AND as one all-reduce after backward, with
      allreduce_exposed_us for both;
  config.fp32_emulated_bf16x9 / _bf16x6 (N = 1) = the headline workload in the two fp32-emulation modes (same parity test as
      the headline; reported beside it, never instead of it);
  config.dropin_images_per_s (N = 1) = the route INTEGRATION.md §A describes: the reference loop
      (critic_vae_amd.train.train: autograd Functions + torch.optim.Adam) fed by FrameFeeder from a pinned
      uint8 host queue, HIP pre-processing and HIP critic included.

  roofline     — the dominant conv kernel of the step, timed LIVE inside the timed region by a HIP event pair
                 recorded around each of its launches on the stream it runs on (cvae_probe_*), priced at its
                 algorithmic FLOPs against the MFMA peak of the dtype it runs on;
  roofline_hbm — the same for the largest HBM-bound kernel, priced at its algorithmic bytes against 8 TB/s;
  traffic      — HBM bytes per launch of that kernel from the committed rocprofv3 PMC passes of the same workload
                 (profiles/traffic_per_launch.json, stamped with the sha256 of the kernel sources; null if stale);
  cpu_baseline — the oracle's training step timed on the host cores (bounded sample).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# this pool's hosts only support dmabuf IPC: without it RCCL's hipIpcGetMemHandle fails at init.  Set before torch / HIP
# load so the documented torchrun form works as well as the self-launch (an exported value wins)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

FLOP_PER_IMG = {64: 1.43762e9, 128: 5.75049e9}   # fwd + dgrad + wgrad of 9 convs + 3 linears
ELEMS_PER_IMG = {64: 2197957, 128: 8790469}      # algorithmic HBM elements per image (fused-ideal)
WEIGHT_BYTES_PER_STEP = 103e6                    # params read x2, grads written, Adam 4R+3W
PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md, matrix fp32
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md, dense bf16
PEAK_HBM_GBPS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (about 6.3 TB/s achievable)
LAYERS = [(3, 32, 64, 0), (32, 64, 32, 0), (64, 128, 16, 0), (128, 256, 8, 0),
          (256, 128, 4, 0), (128, 64, 8, 1), (64, 32, 16, 1), (32, 32, 32, 1), (32, 3, 64, 1)]
KINDS = ("conv_fwd", "conv_dgrad", "conv_wgrad")
MFMA_PROBE_IDS = [k * 9 + l for k in range(3) for l in range(1, 8)]
# HBM-side kernels the library can bracket (include/cvae.h, cvae_probe_config)
HBM_PROBE_NAMES = {0: "e1_fwd", 8: "d4_fwd", 17: "d4_bwd", 18: "e1_wgrad", 27: "bn_pool_bwd_apply_L0",
                   28: "bn_pool_bwd_apply_L1", 29: "bn_pool_bwd_apply_L2", 30: "bn_pool_bwd_apply_L3",
                   31: "msssim_fwd_level0"}
HBM_PROBE_IDS = sorted(HBM_PROBE_NAMES)
BN_CH_H = [(32, 64), (64, 32), (128, 16), (256, 8)]
TRAFFIC_JSON = os.path.join(ROOT, "profiles", "traffic_per_launch.json")

PRESETS = {   # BASELINE.json configs (configs[0] is the CPU case: cpu_baseline.batch32)
    "config1": dict(precision="f32", batch=256, width=64, label="BASELINE.json configs[1]: fp32"),
    "config2": dict(precision="bf16", batch=2048, width=64, label="BASELINE.json configs[2]: bf16-MFMA"),
    "config4": dict(precision="bf16", batch=2048, width=64, label="BASELINE.json configs[3] per-GPU shard (2048 of a global 2048*N): bf16-MFMA"),
    "config5": dict(precision="bf16", batch=1024, width=128, label="BASELINE.json configs[4] per-GPU shard (1024 of a global 1024*N), 128x128 frames: bf16-MFMA"),
}
DTYPE_NOTE = {
    "f32": "f32",
    "bf16": "bf16 (MFMA operands of every conv pass incl. E1/D4; activations and activation gradients stored as bf16; "
            "fp32 accumulate, BatchNorm statistics, loss, master weights, gradients, Adam).  NOT held to the 1e-4 bar: "
            "outputs within 3e-2, loss within 2e-3, gradient tensors 1e-4..0.09 relative L2 of the fp32 oracle "
            "(first conv's weight 0.10 at 64x64 / 0.16 at 128x128), bounds asserted in tests/test_gpu_bf16.py",
    "bf16x9": "f32 emulated: 3-way exact bf16 operand splits, 9 bf16 MFMAs per product block, f32 elsewhere",
    "bf16x6": "f32 emulated: 3-way exact bf16 operand splits, the 6 leading partial products, f32 elsewhere"}


def conv_flops(layer, B, width=64):
    """Algorithmic FLOPs of one pass (forward, dgrad or wgrad) of conv `layer` on B images."""
    cin, cout, h, _ = LAYERS[layer]
    h = h * width // 64
    return 2.0 * 25 * cin * cout * h * h * B


def conv_bytes(pid, B, width, precision):
    """Algorithmic HBM bytes of one launch of conv kernel `pid` = kind*9 + layer (kind 0 forward, 1 input gradient,
    2 weight gradient): each activation tensor it touches once, in the storage type of the mode, plus the fp32 weights
    (read by forward / input gradient, written by the weight gradient).  Layers 5..7 read / write their input side at
    the stored (pre-Upsample) resolution; their input gradient also reads the producer's output for the ReLU mask."""
    kind, layer = pid // 9, pid % 9
    cin, cout, h, up = LAYERS[layer]
    h = h * width // 64
    hin = h // 2 if up else h
    e = 2.0 if precision == "bf16" else 4.0
    a_in, a_out, w = B * hin * hin * cin * e, B * h * h * cout * e, 25.0 * cin * cout * 4
    if kind == 1 and up:
        a_in *= 2                                   # d_in written + the forward activation read for the ReLU mask
    return a_in + a_out + w


def flat_summary(prefix, r):
    """Scalar keys for `config` (the driver's record keeps scalars only): rate, time, both rooflines of a sub-workload."""
    out = {f"{prefix}_images_per_s": r["value"], f"{prefix}_ms_per_step": r["ms_per_step"], f"{prefix}_steps": r["steps"],
           f"{prefix}_warmup": r["warmup"], f"{prefix}_batch_per_gpu": r["batch_per_gpu"], f"{prefix}_final_loss": r["final_loss"]}
    for k in ("whole_step_algorithmic_TFLOPs", "whole_step_algorithmic_GBps"):
        if k in r:
            out[f"{prefix}_{k}"] = r[k]
    rf, rh = r.get("roofline"), r.get("roofline_hbm")
    if rf:
        out.update({f"{prefix}_roofline_kernel": rf["kernel"], f"{prefix}_roofline_frac": rf["frac"],
                    f"{prefix}_roofline_achieved_TFLOPs": rf["achieved"], f"{prefix}_roofline_peak_TFLOPs": rf["peak"],
                    f"{prefix}_roofline_avg_launch_us": rf["avg_launch_us"], f"{prefix}_roofline_launches_timed": rf["launches_timed"],
                    f"{prefix}_roofline_traffic_bytes": rf["traffic"],
                    f"{prefix}_roofline_algorithmic_bytes": rf["algorithmic_bytes_per_launch"],
                    f"{prefix}_roofline_traffic_over_algorithmic": (None if not rf["traffic"] else
                                                                    round(rf["traffic"] / rf["algorithmic_bytes_per_launch"], 3))})
    if rh:
        out.update({f"{prefix}_roofline_hbm_kernel": rh["kernel"], f"{prefix}_roofline_hbm_frac": rh["frac"],
                    f"{prefix}_roofline_hbm_achieved_GBps": rh["achieved"], f"{prefix}_roofline_hbm_avg_launch_us": rh["avg_launch_us"],
                    f"{prefix}_roofline_hbm_traffic_over_algorithmic": (None if not rh["traffic"] else
                                                                        round(rh["traffic"] / rh["algorithmic_bytes_per_launch"], 3))})
    return out


def compact(r):
    """A sub-workload's nested object without the per-kernel tables and the long description strings (those go to the
    detail file): what stays is what the flat keys summarise, in structured form."""
    r = dict(r)
    for k in ("workload", "dtype", "frame", "n_gpus"):
        r.pop(k, None)
    for k in ("roofline", "roofline_hbm"):
        if k in r:
            r[k] = {a: b for a, b in r[k].items() if not a.startswith("in_step_")}
    return r


def probe_name(pid):
    return HBM_PROBE_NAMES.get(pid) or f"{KINDS[pid // 9]}_L{pid % 9}"


def hbm_kernel_bytes(pid, B, width, precision, two_pass_e1=True):
    """Algorithmic HBM bytes of one launch of an HBM-side kernel (each tensor read / written once)."""
    e = 2.0 if precision == "bf16" else 4.0        # activation element size
    W2 = float(width * width)
    x_b, half32 = 3 * W2 * 4.0, (W2 / 4) * 32 * e
    xp_b = W2 * 8.0                                # bf16 mode: the frame as packed bf16 (r, g, b, 0) pixels, written by E1's statistics pass (round 5)
    if pid in (27, 28, 29, 30):                    # BatchNorm + pool backward apply: read y, a, da; write dy
        c, h = BN_CH_H[pid - 27]
        h = h * width // 64
        return 2.5 * B * h * h * c * e
    if pid == 0:                                   # E1 forward: fp32 = read x, write y0; bf16 (second pass) = read the packed frame, write a0
        return B * ((xp_b + half32) if precision == "bf16" else (x_b + W2 * 32 * e))
    if pid == 18:                                  # E1 weight gradient with block 0's BatchNorm/pool backward fused:
        return B * ((xp_b if precision == "bf16" else x_b + W2 * 32 * e) + 2 * half32)    # x (bf16: its packed copy), a0, d_a0 (+ y0 in fp32 mode)
    if pid == 8:                                   # D4 forward: read o3, write recon
        return B * (half32 + x_b)
    if pid == 17:                                  # D4 backward: read d_recon, recon (or dOut) and o3, write d_o3
        return B * ((2 * x_b if precision == "bf16" else x_b) + 2 * half32)
    if pid == 31:                                  # MS-SSIM level 0: read x, recon; write F + two quarter-size images
        return B * (2 * x_b + x_b + 2 * x_b / 4)
    raise KeyError(pid)


def measured_traffic(workload_key, name):
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
        return j["workloads"][workload_key]["bytes_per_launch"].get(name)
    except (OSError, ValueError, KeyError, ImportError):
        return None


def cpu_baseline(B, steps=None, width=64):
    """The oracle's training step (same ATen CPU kernels as the reference, equality pinned by
    tests/golden) timed on this host: bounded sample of the same workload."""
    from critic_vae_amd import synth
    from oracle import cvae_oracle as orc
    if steps is None:                                   # about 10-30 s of CPU work whatever the batch
        steps = max(1, min(10, 2560 * 64 * 64 // (B * width * width)))        # B = 256: 10 steps = 2560 images, ~12 s on 16 cores
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


def self_launch(n):
    """`python bench.py --gpus N` outside torchrun: start N ranks as CHILD processes (one per GPU) through
    torch.distributed.run and pass their output and exit code through.  Called before anything in this
    process initialises HIP — a process that has touched the GPU is never exec'd or re-used.  torchrun picks
    the rendezvous port itself (--standalone: no bind-then-close race)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={n}", os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


class Ctx:
    def __init__(self, world, rank, dev):
        self.world, self.rank, self.dev = world, rank, dev

    def note(self, msg):
        if self.rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    def barrier(self):
        if self.world > 1:
            torch.distributed.barrier()


def time_workload(cx, spec, steps, warmup, probing, allreduce_dtype=None, overlaps=(None,), fwd_bwd_rate=False, side_stream_wgrad=False):
    """Build the model for `spec` (precision, batch, width), run `warmup` untimed + `steps` timed training steps
    bracketed by barrier + synchronize on both sides, max over ranks.  Returns (result dict, trainer, inputs).
    overlaps: all-reduce modes to time at N > 1 (None = the trainer's default); the first one is the result, the
    others are returned under result["allreduce_modes"]."""
    from critic_vae_amd import dp
    from critic_vae_amd.nets import VariationalAutoencoder
    from critic_vae_amd.train import FusedTrainer

    prec, B, Wd = spec["precision"], spec["batch"], spec["width"]
    world, rank, dev = cx.world, cx.rank, cx.dev
    vae = VariationalAutoencoder(width=Wd, max_batch=B, seed=0, precision=prec, overlap_wgrad=side_stream_wgrad).to(dev)
    tr = FusedTrainer(vae, world_size=world, reduce_dtype=allreduce_dtype)
    tr.measure_exposed = world > 1
    H = vae.handle
    # synthetic inputs, resident in HBM before the timed region; each rank its own shard
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    pool = [(torch.rand(B, 3, Wd, Wd, device=dev, generator=gen), torch.rand(B, 1, device=dev, generator=gen),
             torch.randn(B, 32, device=dev, generator=gen)) for _ in range(4)]
    cx.note(f"{spec.get('key', 'workload')}: {prec}, batch {B}/GPU, {Wd}x{Wd}, world {world}")

    def timed(overlap):
        if overlap is not None:
            tr.overlap = bool(overlap) and world > 1
        survey, dominant, dominant_hbm = {}, None, None
        n_survey = min(3, warmup) if probing else 0
        for i in range(warmup):
            if probing and i == warmup - n_survey:
                torch.cuda.synchronize()
                H.probe_config(MFMA_PROBE_IDS + HBM_PROBE_IDS)   # last warm-up steps: time every probed kernel in-step
            tr.step(*pool[i % len(pool)])
        torch.cuda.synchronize()
        if probing and n_survey:
            for pid in MFMA_PROBE_IDS + HBM_PROBE_IDS:
                ms = H.probe_read(pid)
                if ms:
                    survey[pid] = sum(ms) / len(ms)
            dominant = max((p for p in survey if p in MFMA_PROBE_IDS), key=survey.get)
            # HBM-bound = algorithmic bytes / 6.3 TB/s exceed algorithmic FLOPs / MFMA peak: the BatchNorm apply passes always;
            # E1 / D4 only in bf16 mode (on the fp32 MFMA their 75-tap GEMMs take longer than their bytes); MS-SSIM level 0
            # is VALU-bound: listed, never ranked
            hbm_ids = HBM_PROBE_IDS if prec == "bf16" else [27, 28, 29, 30]
            hbm_class = [p for p in survey if p in hbm_ids and p != 31]
            dominant_hbm = max(hbm_class, key=survey.get) if hbm_class else None
            H.probe_config([dominant] + ([dominant_hbm] if dominant_hbm is not None else []))
            cx.note(f"dominant kernel {probe_name(dominant)} ({survey[dominant] * 1e3:.1f} us)")
        tr.exposed_us()                       # drop the warm-up samples
        cx.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            scal = tr.step(*pool[i % len(pool)])
        torch.cuda.synchronize()
        cx.barrier()
        torch.cuda.synchronize()
        dt = dp.max_over_ranks(time.perf_counter() - t0, dev)
        loss = float(scal[0].item())
        exposed = tr.exposed_us()
        cx.note(f"timed {steps} steps in {dt:.3f}s, loss {loss}")
        r = {"value": round(world * B * steps / dt, 1), "unit": "images/s", "ms_per_step": round(dt / steps * 1e3, 4),
             "steps": steps, "warmup": warmup, "dtype": DTYPE_NOTE[prec],
             "workload": f"{spec['label']} train step (fwd+MS-SSIM/KLD loss+bwd+Adam), batch {B}/GPU, {Wd}x{Wd}x3 frames + critic scalars",
             "batch_per_gpu": B, "global_batch": world * B, "frame": f"{Wd}x{Wd}x3", "n_gpus": world,
             "final_loss": loss, "loss_finite": bool(loss == loss and abs(loss) != float("inf"))}
        if world > 1:
            r["grad_allreduce"] = "3 buckets overlapped with backward" if tr.overlap else "single, after backward"
            r["allreduce_exposed_us"] = None if exposed is None else round(exposed, 1)
        if rank == 0 and world == 1:
            img_s = B * steps / dt
            ebytes = 2.0 if prec == "bf16" else 4.0
            r["whole_step_algorithmic_TFLOPs"] = round(img_s * FLOP_PER_IMG[Wd] / 1e12, 2)
            r["whole_step_algorithmic_GBps"] = round((ELEMS_PER_IMG[Wd] * ebytes * B + WEIGHT_BYTES_PER_STEP) * steps / dt / 1e9, 1)
            if dominant is not None:
                wkey = f"{prec}_b{B}_w{Wd}"
                ms = H.probe_read(dominant)
                ms_hbm = H.probe_read(dominant_hbm) if dominant_hbm is not None else []
                H.probe_config([])
                sec = sum(ms) / len(ms) * 1e-3
                fl = conv_flops(dominant % 9, B, Wd)
                # fp32 emulation: forward / dgrad of E2..E4, D0 run 9 (or 6) bf16 MFMAs per fp32 product block -> peak / 9 (/ 6);
                # so do their weight gradients (round 4, conv_wgrad_split.hip); every other kernel stays on the fp32 MFMA
                emu = prec in ("bf16x9", "bf16x6") and 1 <= dominant % 9 <= 4
                peak = PEAK_BF16_MFMA_TFLOPS if prec == "bf16" else (PEAK_BF16_MFMA_TFLOPS / (9 if prec == "bf16x9" else 6) if emu
                                                                    else PEAK_FP32_MFMA_TFLOPS)
                r["roofline"] = {
                    "bound": "mfma", "kernel": probe_name(dominant), "achieved": round(fl / sec / 1e12, 2),
                    "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(fl / sec / 1e12 / peak, 4),
                    "traffic": measured_traffic(wkey, probe_name(dominant)),
                    "avg_launch_us": round(sec * 1e6, 2), "launches_timed": len(ms),
                    "algorithmic_flops_per_launch": fl, "algorithmic_bytes_per_launch": conv_bytes(dominant, B, Wd, prec),
                    "in_step_TFLOPs_all_conv_kernels": {probe_name(k): round(conv_flops(k % 9, B, Wd) / (v * 1e-3) / 1e12, 1)
                                                        for k, v in sorted(survey.items()) if k in MFMA_PROBE_IDS}}
                if ms_hbm:          # second roofline: the largest HBM-bound kernel of the step
                    sh = sum(ms_hbm) / len(ms_hbm) * 1e-3
                    by = hbm_kernel_bytes(dominant_hbm, B, Wd, prec)
                    r["roofline_hbm"] = {
                        "bound": "hbm", "kernel": probe_name(dominant_hbm), "achieved": round(by / sh / 1e9, 1),
                        "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": round(by / sh / 1e9 / PEAK_HBM_GBPS, 4),
                        "traffic": measured_traffic(wkey, probe_name(dominant_hbm)),
                        "avg_launch_us": round(sh * 1e6, 2), "launches_timed": len(ms_hbm),
                        "algorithmic_bytes_per_launch": by,
                        "in_step_us_GBps_all_hbm_side_kernels": {
                            probe_name(k): [round(v * 1e3, 1), round(hbm_kernel_bytes(k, B, Wd, prec) / (v * 1e-3) / 1e9, 0)]
                            for k, v in sorted(survey.items()) if k in HBM_PROBE_IDS}}
        return r

    res = timed(overlaps[0])
    if len(overlaps) > 1:
        res["allreduce_modes"] = {("overlap" if o else "single"): {k: v for k, v in timed(o).items() if k in
                                  ("value", "ms_per_step", "grad_allreduce", "allreduce_exposed_us", "final_loss")} for o in overlaps[1:]}
    if world > 1:
        # what the collectives really ran on: ranks counted by an all-reduce, backend, device of every rank
        ones = torch.ones(1, device=dev)
        torch.distributed.all_reduce(ones)
        mine = torch.zeros(world, dtype=torch.int64, device=dev)
        mine[rank] = dev.index
        torch.distributed.all_reduce(mine)
        res["distributed"] = {"backend": torch.distributed.get_backend(), "ranks_counted_by_allreduce": int(ones.item()),
                              "world_size": torch.distributed.get_world_size(), "rank_devices": [int(v) for v in mine.tolist()],
                              "allreduce_exposed_us": res.get("allreduce_exposed_us"),
                              "allreduce_dtype": tr.reduce_dtype,
                              "allreduce_bytes": int(tr.grads.numel() * (2 if tr.reduce_dtype == "bf16" else 4))}
    if fwd_bwd_rate:
        # SURVEY 8d also asks for the C-ABI path proper (forward + loss + backward, no all-reduce / optimizer):
        # same rotating inputs as the timed region, 3 warm-up passes, at least as many passes as timed steps
        theta = vae.theta.data

        def fwd_loss_bwd(i):
            x0, p0, e0 = pool[i % len(pool)]
            H.forward(B, x0, p0, e0, theta, vae.bn_state, tr.mu, tr.logvar, tr.recon, tr.ws, train=True)
            H.loss(B, x0, tr.mu, tr.logvar, tr.recon, tr.ws, tr.scalars, tr.d_recon, tr.d_mu, tr.d_logvar)
            H.backward(B, x0, p0, e0, theta, tr.logvar, tr.recon, tr.d_recon, tr.d_mu, tr.d_logvar, tr.ws, tr.grads)

        n_fb = max(steps, 20)
        for i in range(3):
            fwd_loss_bwd(i)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(n_fb):
            fwd_loss_bwd(i)
        torch.cuda.synchronize()
        res["fwd_loss_bwd_only_images_per_s_per_gpu"] = round(B * n_fb / (time.perf_counter() - t1), 1)
    del tr, vae, pool
    torch.cuda.empty_cache()
    return res


def dropin_rate(cx, B=256, steps=40, warmup=8):
    """critic_vae_amd.train.train (the reference loop: autograd Functions + torch.optim.Adam)
    over a host uint8 dataset through FrameFeeder (pinned double buffer, H2D on a side stream) with the HIP
    pre-processing and the HIP critic in the loop.  One epoch of warmup+steps batches; the first `warmup`
    batches are not timed."""
    import numpy as np
    from critic_vae_amd import synth
    from critic_vae_amd import train as T
    from critic_vae_amd.critic import Critic
    from critic_vae_amd.nets import VariationalAutoencoder

    dev = cx.dev
    vae = VariationalAutoencoder(max_batch=B, seed=0).to(dev)
    critic = Critic(handle=vae.handle).to(dev)
    critic.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_critic_params(0).items()})
    rng = np.random.default_rng(7)
    frames = rng.integers(0, 256, size=((warmup + steps) * B, 64, 64, 3), dtype=np.uint8)
    marks = []

    class _Clock:                      # train() logs every log_n images: use the log hook as the warm-up / end marker
        def __call__(self, msg):
            torch.cuda.synchronize()
            marks.append(time.perf_counter())

    np.random.seed(0)
    T.train(vae, frames, critic, dev, epochs=1, batch_size=B, log_n=warmup * B, log=_Clock())
    torch.cuda.synchronize()
    t_end = time.perf_counter()
    # log fires at batch_i = 0 (before warm-up) and batch_i = warmup*B (start of the timed part; the sync there drains
    # the warm-up batches), then every warmup*B images: the timed part is from marks[1] to the end of the epoch
    t_start = marks[1]
    n_timed = (steps - 1) * B          # batch `warmup` itself was enqueued and drained before marks[1] was taken
    del vae, critic
    torch.cuda.empty_cache()
    return {"value": round(n_timed / (t_end - t_start), 1), "unit": "images/s",
            "what": f"critic_vae_amd.train.train (reference loop vae.py:33-66: autograd + torch.optim.Adam on the flat parameter), "
                    f"batch {B}, fp32, {steps - 1} timed batches after {warmup + 1}; frames start as uint8 HWC on the HOST: pinned "
                    f"double-buffered gather, H2D on a side stream, cvae_preprocess_u8 + cvae_critic_forward on the GPU "
                    f"(PCIe-inclusive; log hook synchronises every {warmup} batches)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default 256 = BASELINE.json configs[1])")
    ap.add_argument("--width", type=int, choices=[64, 128], default=None, help="frame size (128: BASELINE configs[4] shape)")
    ap.add_argument("--precision", choices=["f32", "bf16", "bf16x9", "bf16x6"], default=None,
                    help="f32 = the 1e-4-parity path (default, BASELINE configs[1]); bf16 = bf16-MFMA mode (configs[2..4])")
    ap.add_argument("--preset", choices=sorted(PRESETS), default=None,
                    help="make a BASELINE.json config the HEADLINE workload (sets precision/batch/width; conflicts with those flags)")
    ap.add_argument("--allreduce-dtype", choices=["f32", "bf16"], default=None,
                    help="wire format of the gradient all-reduce at N > 1 (default f32; bf16 halves the bytes, optional per SURVEY 8e)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-probe", action="store_true")
    ap.add_argument("--no-fwd-bwd-rate", action="store_true", help="skip the extra forward+loss+backward-only loop (profiling runs)")
    ap.add_argument("--no-extra-configs", action="store_true", help="headline workload only (profiling runs)")
    args = ap.parse_args()
    explicit = args.preset or args.batch is not None or args.width is not None or args.precision is not None
    if args.preset:
        clash = [k for k in ("batch", "width", "precision") if getattr(args, k) is not None and getattr(args, k) != PRESETS[args.preset][k]]
        if clash:
            ap.error(f"--preset {args.preset} fixes {', '.join('--' + k for k in clash)}; drop the conflicting option(s)")
        head = dict(PRESETS[args.preset], key=args.preset)
    else:
        prec, B, Wd = args.precision or "f32", args.batch or 256, args.width or 64
        label = {"f32": "BASELINE.json configs[1]: fp32" if (B, Wd) == (256, 64) else "fp32",
                 "bf16": "bf16-MFMA", "bf16x9": "fp32 emulated by 3-way bf16 splits (9 products)",
                 "bf16x6": "fp32 emulated by 3-way bf16 splits (6 products)"}[prec]
        head = dict(precision=prec, batch=B, width=Wd, label=label, key="config1" if not explicit else "custom")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))

    from critic_vae_amd import dp

    world, rank, local = dp.init()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs MI355X GPUs (the HIP library has no CPU fallback)")
    torch.cuda.set_device(dp.device_index(local))
    dev = torch.device("cuda", dp.device_index(local))
    cx = Ctx(world, rank, dev)
    probing = world == 1 and not args.no_probe

    h = time_workload(cx, head, args.steps, args.warmup, probing, args.allreduce_dtype,
                      fwd_bwd_rate=not args.no_fwd_bwd_rate)
    res = {
        "metric": f"VAE train-step images/sec on {head['width']}x{head['width']}x3 frames",
        "value": h["value"], "unit": "images/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": h["ms_per_step"], "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": h["dtype"], "data": "synthetic",
        "config": {"workload": h["workload"], "preset": head["key"], "global_batch": h["global_batch"], "frame": h["frame"],
                   "parallelism": f"dp{world}", "optimizer": "fused flat Adam",
                   "grad_allreduce": h.get("grad_allreduce", "none"),
                   "final_loss": h["final_loss"], "loss_finite": h["loss_finite"]},
    }
    for k in ("whole_step_algorithmic_TFLOPs", "whole_step_algorithmic_GBps", "fwd_loss_bwd_only_images_per_s_per_gpu"):
        if k in h:
            res["config"][k] = h[k]
    for k in ("roofline", "roofline_hbm"):
        if k in h:
            res[k] = h[k]
    if "distributed" in h:
        res["config"]["distributed"] = h["distributed"]
        res["rccl_ranks"] = h["distributed"]["ranks_counted_by_allreduce"] if h["distributed"]["backend"] == "nccl" else 0
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(head["batch"], width=head["width"])
        if head["batch"] != 32:       # the reference's own CPU-runnable case (configs[0], batch 32) beside it
            res["cpu_baseline"]["batch32"] = cpu_baseline(32, steps=10, width=head["width"])

    # The other BASELINE configs, in the same process and with the same timed loop, after the headline.  Each lands in
    # `config` twice: as FLAT scalar keys (config2_images_per_s, config2_roofline_frac, ... — the driver's record keeps
    # scalars only) and as a compact nested object; the per-kernel tables and description strings of every sub-workload go
    # to the detail file named by config.detail_file (and to stderr), so the stdout line stays short enough to be kept whole.
    detail = {"headline": h}
    cfgd = res["config"]
    if not explicit and not args.no_extra_configs:
        if world == 1:
            for key in ("config2", "config5"):
                full = time_workload(cx, dict(PRESETS[key], key=key), args.steps, args.warmup, probing)
                # the same workload with the weight-gradient kernels on the library's side stream (cvae_config.overlap_wgrad,
                # bit-identical results): kernels then share the chip, so it is a rate beside the line, without rooflines
                ss = time_workload(cx, dict(PRESETS[key], key=key + " + side-stream weight gradients"), args.steps, args.warmup,
                                   False, side_stream_wgrad=True)
                full["side_stream_wgrad"] = {k: ss[k] for k in ("value", "unit", "ms_per_step", "steps", "final_loss")}
                detail[key] = full
                cfgd.update(flat_summary(key, full))
                cfgd[f"{key}_side_stream_wgrad_images_per_s"] = ss["value"]
                cfgd[f"{key}_dtype"] = "bf16 MFMA + bf16 activation storage; not held to 1e-4: gradient tensors 1e-4..0.16 rel-L2 of fp32 (tests/test_gpu_bf16.py)"
                cfgd[key] = compact(full)
            # the same configs[1] workload in the two fp32-EMULATION modes (exact 3-way bf16 operand splits on the bf16 MFMA,
            # DESIGN.md 7b): they pass the same decisions-imposed 1e-4 parity test as the fp32 headline
            # (tests/test_gpu_step.py::test_step_b256_fp32_against_oracle[bf16x9|bf16x6]) but are reported beside it, never as it
            for prec, what in (("bf16x9", "all nine partial products: exact products, fp32 accumulate"),
                               ("bf16x6", "the six leading partial products (drops <= 3*2^-24 of each product)")):
                spec = dict(PRESETS["config1"], precision=prec, key=f"fp32_emulated_{prec}",
                            label=f"BASELINE.json configs[1] workload with fp32 EMULATED on the bf16 MFMA ({what}; forward + input gradients of E2-E4 / D0)")
                full = time_workload(cx, spec, args.steps, args.warmup, probing)
                detail[f"fp32_emulated_{prec}"] = full
                keep = ("images_per_s", "ms_per_step", "steps", "final_loss", "roofline_kernel", "roofline_frac", "roofline_avg_launch_us")
                cfgd.update({k: v for k, v in flat_summary(prec, full).items() if k[len(prec) + 1:] in keep})       # nested form: detail file only
            dr = dropin_rate(cx)
            detail["dropin"] = dr
            cfgd["dropin_images_per_s"] = dr["value"]
            # the step the reference runs by default (vae_parameters.py:10, batch_size = 128): fp32, one number, no rooflines
            b128 = time_workload(cx, dict(PRESETS["config1"], batch=128, key="b128", label="the reference's default batch (vae_parameters.py:10): fp32"),
                                 max(20, min(args.steps, 50)), min(args.warmup, 10), False)
            detail["b128"] = b128
            cfgd["b128_images_per_s"] = b128["value"]
        else:
            for key in ("config4", "config5"):
                full = time_workload(cx, dict(PRESETS[key], key=key), args.steps, args.warmup, False,
                                     args.allreduce_dtype, overlaps=(True, False))
                detail[key] = full
                cfgd.update(flat_summary(key, full))
                for mode, m in full.get("allreduce_modes", {}).items():
                    cfgd[f"{key}_allreduce_{mode}_images_per_s"] = m["value"]
                    cfgd[f"{key}_allreduce_{mode}_exposed_us"] = m.get("allreduce_exposed_us")
                cfgd[f"{key}_allreduce_exposed_us"] = full.get("allreduce_exposed_us")
                cfgd[key] = compact(full)
    # Key order of `config`: the driver's record keeps the first 24 scalar keys — those are the ones a reader needs to check every config
    # (headline bookkeeping, then rate / time / both roofline fractions of config2 and config5, then the emulation, drop-in and default-batch
    # rates); everything else follows (and is whole in the stdout line and the detail file).
    first = ["workload", "preset", "global_batch", "final_loss", "loss_finite", "fwd_loss_bwd_only_images_per_s_per_gpu"]
    for key in ("config2", "config5") if world == 1 else ("config4", "config5"):
        first += [f"{key}_{k}" for k in ("images_per_s", "ms_per_step", "steps", "roofline_kernel", "roofline_frac",
                                         "roofline_traffic_over_algorithmic", "roofline_hbm_frac")]
    first += ["bf16x9_images_per_s", "bf16x6_images_per_s", "dropin_images_per_s", "b128_images_per_s"]
    res["config"] = cfgd = {**{k: cfgd[k] for k in first if k in cfgd}, **{k: v for k, v in cfgd.items() if k not in first}}
    # the headline keeps its per-kernel tables in the detail file too; the line carries the two rooflines without them
    for k in ("roofline", "roofline_hbm"):
        if k in res:
            res[k] = {a: b for a, b in res[k].items() if not a.startswith("in_step_")}

    if rank == 0:
        path = os.environ.get("CVAE_BENCH_DETAIL") or os.path.join(ROOT, "gpurun_out" if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else "",
                                                                  "bench_detail.json")
        try:
            with open(path, "w") as f:
                json.dump({"line": res, "detail": detail}, f)
            res["config"]["detail_file"] = os.path.relpath(path, ROOT)
        except OSError:
            res["config"]["detail_file"] = None
        print("[bench] detail " + json.dumps(detail), file=sys.stderr, flush=True)
        print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
