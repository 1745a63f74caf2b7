"""sha256 over the kernel sources (critic-vae_amd/csrc/*.hip, *.h, sorted by name): profiles/traffic_per_launch.json
records it, bench.py recomputes it and only reports `roofline.traffic` when the two agree — a PMC figure measured
on other kernels than the ones being timed is never printed."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_sha(root=ROOT):
    h = hashlib.sha256()
    d = os.path.join(root, "critic-vae_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h"))):
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()


if __name__ == "__main__":
    print(kernel_source_sha())
