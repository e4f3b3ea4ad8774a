#!/usr/bin/env python3
"""Where the SGPR spills of a kernel sit (CPU only): per basic block of the gfx950
assembly, the vector / lane (v_readlane, v_writelane = SGPR spill traffic) / scalar-load
instruction counts, with the loop nesting derived from backward branches.

usage: asm_stats.py FILE.s 'demangled substring' [--blocks]
"""

from __future__ import annotations

import re
import subprocess
import sys


def functions(path):
    """name -> list of lines, for every kernel / function body in the .s file."""
    out, cur, name = {}, None, None
    with open(path) as f:
        for line in f:
            m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
            if m:
                name, cur = m.group(1), []
                out[name] = cur
                continue
            if cur is not None:
                if line.startswith(".Lfunc_end"):
                    cur, name = None, None
                    continue
                cur.append(line.rstrip("\n"))
    return out


def analyse(lines, show_blocks=False):
    blocks, order = {}, []
    cur = "entry"
    blocks[cur] = []
    order.append(cur)
    for ln in lines:
        m = re.match(r"^(\.LBB\w+):", ln)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            order.append(cur)
            continue
        s = ln.strip()
        if not s or s.startswith(";") or s.startswith("."):
            continue
        blocks[cur].append(s)
    idx = {b: i for i, b in enumerate(order)}
    # loops: a branch from block i to block j <= i makes [j, i] a loop body
    depth = [0] * len(order)
    loops = []
    for b in order:
        for ins in blocks[b]:
            m = re.match(r"s_c?branch\w*\s+(\.LBB\w+)", ins)
            if m and m.group(1) in idx and idx[m.group(1)] <= idx[b]:
                loops.append((idx[m.group(1)], idx[b]))
    for lo, hi in loops:
        for i in range(lo, hi + 1):
            depth[i] += 1
    tot = {}
    for b in order:
        d = depth[idx[b]]
        t = tot.setdefault(d, {"valu": 0, "lane": 0, "smem": 0, "salu": 0, "vmem": 0, "lds": 0,
                               "trans": 0, "f64": 0})
        for ins in blocks[b]:
            op = ins.split()[0]
            if op.startswith("v_readlane") or op.startswith("v_writelane"):
                t["lane"] += 1
                t["valu"] += 1
            elif op.startswith("v_"):
                t["valu"] += 1
                if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)", op):
                    t["trans"] += 1
                if op.endswith("_f64") or "_f64_" in op:
                    t["f64"] += 1
            elif op.startswith("s_load") or op.startswith("s_buffer_load"):
                t["smem"] += 1
            elif op.startswith("s_"):
                t["salu"] += 1
            elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") \
                    or op.startswith("scratch_"):
                t["vmem"] += 1
            elif op.startswith("ds_"):
                t["lds"] += 1
        if show_blocks:
            c = sum(1 for i in blocks[b] if i.startswith("v_"))
            l = sum(1 for i in blocks[b] if i.startswith(("v_readlane", "v_writelane")))
            if c or l:
                print(f"    {b:<14} depth {d}  valu {c:4d}  lane {l:3d}")
    return tot, len(loops)


def main(argv):
    path, pat = argv[0], argv[1]
    show = "--blocks" in argv
    fns = functions(path)
    names = list(fns)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True,
                         text=True).stdout.splitlines()
    for n, d in zip(names, dem):
        d = re.sub(r"\(.*$", "", d.replace("void ", "")).replace("ol::", "")
        if pat not in d:
            continue
        tot, nloops = analyse(fns[n], show)
        print(f"{d}   ({nloops} backward branches)")
        for dep in sorted(tot):
            t = tot[dep]
            print(f"    loop depth {dep}: valu {t['valu']:5d} (lane/spill {t['lane']:4d}, "
                  f"transcendental {t['trans']:3d}, f64 {t['f64']:4d})  smem {t['smem']:4d}  "
                  f"salu {t['salu']:5d}  vmem {t['vmem']:4d}  lds {t['lds']:3d}")


if __name__ == "__main__":
    main(sys.argv[1:])
