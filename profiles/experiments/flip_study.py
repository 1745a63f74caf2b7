"""Where does the fp32 HIP step differ from the oracle at BASELINE config 2's size (B=256)?

Prints, per gradient tensor, the absolute / relative-to-max deviation of the HIP step from (a) the plain
oracle, (b) the oracle with the HIP path's discrete decisions imposed (max-pool argmax, ReLU masks), and
(c) the oracle's own 1-thread vs N-thread deviation; plus the number of decision flips and how close to
a tie each flipped unit is.  Basis of tests/test_gpu_step.py::test_step_b256_fp32_against_oracle.

    python profiles/experiments/flip_study.py [B]
"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from critic_vae_amd import synth                       # noqa: E402
from critic_vae_amd.nets import VariationalAutoencoder  # noqa: E402
from oracle import cvae_oracle as orc                   # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "tests"))
from decisions import hip_decisions, oracle_decisions   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256


def main():
    x, pred, eps = (torch.from_numpy(a) for a in synth.make_batch(1234, 0, B))
    vae = VariationalAutoencoder(max_batch=B, seed=0).cuda()
    vae.load_reference_params(synth.make_params(0))
    out = vae(x.cuda(), pred.cuda(), eps=eps.cuda())
    vae.vae_loss(*out)["total_loss"].backward()
    torch.cuda.synchronize()
    g_hip = {k: v.cpu().double() for k, v in vae.reference_grads().items()}
    d_hip = hip_decisions(vae, B)

    def oracle(decisions=None, taps=None, threads=None):
        if threads:
            torch.set_num_threads(threads)
        p = orc.to_torch(synth.make_params(0), requires_grad=True)
        o = orc.train_step(p, x, pred, eps, bn_state=orc.new_bn_state(p), taps=taps, decisions=decisions)
        return {k: v.grad.double() for k, v in p.items()}, o

    nthr = torch.get_num_threads()
    taps = {}
    g_orc, o = oracle(taps=taps)
    d_orc = oracle_decisions(taps)
    print(f"B={B} threads={nthr} loss hip {float(vae.last_scalars[0]):.7f} oracle {float(o['total_loss']):.7f}")
    total_flips = 0
    for k in d_hip:
        flips = d_hip[k] != d_orc[k]
        nf = int(flips.sum())
        total_flips += nf
        msg = f"  {k:10s} units {flips.numel():10d} flips {nf:5d}"
        if nf and k.startswith("pool"):
            l = int(k[-1])
            n = taps[f"enc_n{l}"].detach()
            a = n.flatten(2).gather(2, d_hip[k].flatten(2)).view(flips.shape)
            b = n.flatten(2).gather(2, d_orc[k].flatten(2)).view(flips.shape)
            msg += f"  max |n[hip argmax]-n[oracle argmax]| at flips {float((a - b).abs()[flips].max()):.3e}"
        elif nf and k.startswith("relu_enc"):
            l = int(k[-1])
            n = taps[f"enc_n{l}"].detach()
            pooled = F.max_pool2d(n, 2)
            msg += f"  max |pooled pre-activation| at flips {float(pooled.abs()[flips].max()):.3e}"
        elif nf:
            i = int(k[-1])
            msg += f"  max |pre-activation| at flips {float(taps[f'dec_pre{i}'].detach().abs()[flips].max()):.3e}"
        print(msg)
    print(f"  total flips {total_flips}")
    g_forced, _ = oracle(decisions=d_hip)
    g_1thr, _ = oracle(threads=1)
    torch.set_num_threads(nthr)
    print(f"{'tensor':38s} {'max|g|':>10s} | {'abs hip-orc':>11s} {'rel':>9s} | {'abs hip-forced':>14s} {'rel':>9s} | {'orc 1thr-Nthr rel':>17s}")
    worst = [0.0, 0.0, 0.0]
    for k in g_orc:
        mx = max(float(g_orc[k].abs().max()), 1e-30)
        e1 = float((g_hip[k] - g_orc[k]).abs().max())
        e2 = float((g_hip[k] - g_forced[k]).abs().max())
        e3 = float((g_1thr[k] - g_orc[k]).abs().max())
        pre_bn_bias = k.startswith("encoder.model.") and k.endswith(".bias") and int(k.split(".")[2]) % 4 == 0
        if not pre_bn_bias:
            worst = [max(worst[0], e1 / mx), max(worst[1], e2 / mx), max(worst[2], e3 / mx)]
        print(f"{k:38s} {mx:10.3e} | {e1:11.3e} {e1 / mx:9.2e} | {e2:14.3e} {e2 / mx:9.2e} | {e3 / mx:17.2e}{'  (pre-BN bias)' if pre_bn_bias else ''}")
    print(f"worst rel (pre-BN biases excluded): hip-vs-oracle {worst[0]:.2e}  hip-vs-forced {worst[1]:.2e}  oracle 1thr-vs-{nthr}thr {worst[2]:.2e}")


if __name__ == "__main__":
    main()
