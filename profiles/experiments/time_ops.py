"""Time single C-ABI ops with HIP events (experiments; not part of the bench):
    python profiles/experiments/time_ops.py msssim [B] [width]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from critic_vae_amd import lib as cvlib  # noqa: E402


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    what = sys.argv[1]
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    W = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    H = cvlib.Handle(W, B)
    dev = torch.device("cuda:0")
    if what == "msssim":
        a = torch.rand(B, 3, W, W, device=dev) * 0.9 + 0.05
        b = (a + 0.1 * torch.rand(B, 3, W, W, device=dev)).clamp(0, 1)
        ws = torch.empty(H.op_msssim_ws_floats(B), device=dev)
        scal = torch.empty(16, device=dev)
        d = torch.empty_like(a)
        t_all = timeit(lambda: H.op_msssim(B, a, b, ws, scal, d))
        if hasattr(H.lib, "cvae_ms_dbg_read"):        # -DMS_TIMING build: phase timestamps (clock64) of waves 0 / 15 of four workgroups
            import ctypes
            buf = (ctypes.c_longlong * 128)()
            torch.cuda.synchronize()
            H.lib.cvae_ms_dbg_read(buf)
            names = ["stage", "B0", "pool+H1", "B1", "V1", "B2", "part+H2", "B3", "V2"]
            for g in range(8):
                t = list(buf[g * 16:g * 16 + 10])
                print(f"wg {1024 * (g // 2)} wave {'0' if g % 2 == 0 else 'last'}: " + "  ".join(f"{n} {t[i + 1] - t[i]}" for i, n in enumerate(names)) + f"  total {t[9] - t[0]}")
        t_fwd = timeit(lambda: H.op_msssim(B, a, b, ws, scal, None))
        print(f"msssim B={B} W={W}: fwd+bwd {t_all:.1f} us, fwd only {t_fwd:.1f} us, loss {float(scal[1]):.6f}")


if __name__ == "__main__":
    main()
