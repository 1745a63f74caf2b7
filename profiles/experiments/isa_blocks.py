"""Static instruction mix per basic block of one kernel in a `hipcc -S` listing.
usage: isa_blocks.py <file.s> <mangled-name substring>"""
import collections
import re
import sys

src = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = next(i for i, l in enumerate(src) if l.startswith("_Z") and key in l and l.rstrip().endswith(tuple(":")) or (l.startswith("_Z") and key in l and ": " in l))
blocks, cur = [], ["entry", collections.Counter()]
for l in src[start + 1:]:
    t = l.strip()
    if t.startswith(".Lfunc_end"):
        break
    if re.match(r"^\.LBB\d+_\d+:", t):
        blocks.append(cur)
        cur = [t.split(":")[0], collections.Counter()]
        continue
    if not t or t.startswith((";", ".")):
        continue
    op = t.split()[0]
    cls = ("pk" if op.startswith("v_pk_") else "mov" if op.startswith(("v_mov", "v_accvgpr")) else "valu" if op.startswith("v_") else
           "ds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "scratch_")) else
           "wait" if op.startswith(("s_waitcnt", "s_barrier", "s_nop")) else "salu")
    cur[1][cls] += 1
    cur[1]["_" + op] += 1
    if op.startswith(("s_cbranch", "s_branch")):
        cur[1]["->" + t.split()[-1]] += 1
blocks.append(cur)
for name, c in blocks:
    n = sum(v for k, v in c.items() if not k.startswith(("_", "->")))
    if n < 8:
        continue
    br = [k for k in c if k.startswith("->")]
    print(f"{name:10s} n={n:5d} pk={c['pk']:4d} valu={c['valu']:4d} mov={c['mov']:4d} ds={c['ds']:4d} vmem={c['vmem']:3d} salu={c['salu']:4d} wait={c['wait']:3d} {' '.join(br)}")
    if len(sys.argv) > 3:
        print("    ", ", ".join(f"{k[1:]}:{v}" for k, v in sorted(c.items(), key=lambda kv: -kv[1]) if k.startswith("_"))[:600])
