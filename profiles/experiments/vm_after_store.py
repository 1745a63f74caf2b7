#!/usr/bin/env python3
"""Find "store -> wait" hazards in hipcc -S listings (gfx950).

vmcnt counts loads AND stores of a wave together and retires them in issue order, so an `s_waitcnt vmcnt(N)` that is
reached after global stores were issued waits for those stores' write acknowledgements (thousands of cycles under load)
whenever N is smaller than the number of stores behind the load it really wants.  The classic source shapes: a bias /
mask load whose first use sits behind the previous block's stores, a load issued after a store in an epilogue loop,
stores under a branch (the compiler then cannot count them and emits vmcnt(0)).

    python3 profiles/experiments/vm_after_store.py /tmp/asm/conv_bf16.s [name-filter]

Per kernel, in linear listing order: number of global stores issued before each `s_waitcnt vmcnt(N)` that follows a
store, with the listing line, N and the loads issued since the last store (0 loads since + a wait = a pure store wait).
Loops make the linear view conservative (a wait at a loop top follows the stores of the previous iteration); the
listing line is printed so that each hit can be read in context.
"""
import re
import sys


def kernels(path):
    name, body, start = None, [], 0
    for ln, line in enumerate(open(path), 1):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            if name:
                yield name, start, body
            name, body, start = m.group(1), [], ln
        elif name:
            if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
                yield name, start, body
                name, body = None, []
            else:
                body.append((ln, line.rstrip()))
    if name:
        yield name, start, body


def demangle_hint(n):
    m = re.match(r"_Z\d+(\w+?)I(.*)E", n)
    if not m:
        return n[:80]
    args = re.findall(r"L[ib](\d+)E", m.group(2))
    return f"{m.group(1)}<{','.join(args)}>"


def main():
    args = [a for a in sys.argv[1:] if a != "-a"]         # -a: list the harmless waits (N >= stores issued so far) too
    path = args[0]
    flt = args[1] if len(args) > 1 else ""
    for name, start, body in kernels(path):
        hint = demangle_hint(name)
        if flt and flt not in hint and flt not in name:
            continue
        stores = loads_since = 0
        hits = []
        for ln, line in body:
            s = line.strip()
            if re.match(r"(global|buffer|flat)_store", s) or re.match(r"(global|buffer)_atomic", s):
                stores += 1
                loads_since = 0
            elif re.match(r"(global|buffer|flat)_load", s):
                loads_since += 1
            else:
                m = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", s)
                if m and stores and (int(m.group(1)) < stores or "-a" in sys.argv):
                    hits.append((ln, int(m.group(1)), stores, loads_since))      # N >= stores: no store is among the ops waited for
        if hits:
            print(f"{hint}  (line {start})")
            for ln, n, st, ls in hits:
                print(f"    line {ln}: vmcnt({n}) after {st} stores, {ls} loads since the last store")


if __name__ == "__main__":
    main()
