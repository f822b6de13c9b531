#!/usr/bin/env python3
"""Instruction count of the main loop of a kernel from hipcc's --save-temps assembly.

    tools/isa_loop_count.py FILE.s KERNEL_SUBSTRING [UNROLL]

Finds the kernel whose (mangled) name contains KERNEL_SUBSTRING, takes the LONGEST backward-branch loop in it (label ..
s_cbranch to that label) and prints its instruction mix; with UNROLL (symbols / tiles per loop iteration) also the count
per unit.  Used for profiles/r03_rc_decode_isa.txt (the range decoder is bound by the instruction issue of one wave)."""
import collections
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    unroll = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    lines = open(path).read().splitlines()
    start = None
    for i, ln in enumerate(lines):
        m = re.match(r"^(_Z\w+):", ln)
        if m and key in m.group(1):
            start, name = i, m.group(1)
            break
    if start is None:
        raise SystemExit(f"no kernel matching {key}")
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = lines[start:end + 1]
    labels = {}
    for i, ln in enumerate(body):
        m = re.match(r"^(\.LBB\w+):", ln)
        if m:
            labels[m.group(1)] = i
    best = None
    for i, ln in enumerate(body):
        m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\w+)", ln) or re.match(r"\s+s_branch\s+(\.LBB\w+)", ln)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            span = (labels[m.group(1)], i)
            if best is None or span[1] - span[0] > best[1] - best[0]:
                best = span
    if best is None:
        raise SystemExit("no loop found")
    mix = collections.Counter()
    total = 0
    for ln in body[best[0]:best[1] + 1]:
        t = ln.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        op = t.split()[0]
        total += 1
        cls = ("salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "valu")
        mix[cls] += 1
        mix["op:" + op] += 1
    print(f"kernel {name}")
    print(f"loop lines {best[0]}..{best[1]}: {total} instructions per iteration, {total / unroll:.1f} per unit (unroll {unroll})")
    print("  " + ", ".join(f"{k} {mix[k]} ({mix[k] / unroll:.1f})" for k in ("valu", "salu", "lds", "vmem")))
    ops = sorted(((v, k[3:]) for k, v in mix.items() if k.startswith("op:")), reverse=True)
    print("  " + ", ".join(f"{k} x{v}" for v, k in ops[:24]))


if __name__ == "__main__":
    main()
