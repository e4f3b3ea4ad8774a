#!/usr/bin/env python3
"""Static register budget of every kernel in the trace translation units (CPU only).

hipcc cross-compiles gfx950 device code without a GPU; `-Rpass-analysis=kernel-resource-usage`
reports, per kernel, the VGPR / SGPR allocation, the SGPR and VGPR spill counts, scratch
bytes and the occupancy the allocation allows.  SGPR spills are v_writelane / v_readlane --
VECTOR instructions -- so in the VALU-bound Newton / fp64 kernels they are issue slots the
arithmetic does not get ("What's weak" #2).

usage: kernel_resources.py [--out FILE] [--filter SUBSTR] [-DNAME=VALUE ...]
"""

from __future__ import annotations

import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "optiland_amd", "csrc")
SOURCES = ("trace_kernel_f32.hip", "trace_kernel_f64.hip", "aux_kernels.hip")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on",
         "-fno-math-errno", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage"]
KEYS = ("TotalSGPRs", "VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]",
        "SGPRs Spill", "VGPRs Spill", "LDS Size [bytes/block]")


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names),
                         capture_output=True, text=True, check=True).stdout.splitlines()
    short = []
    for n in out:
        n = re.sub(r"^void ", "", n)
        n = re.sub(r"\(.*$", "", n)  # argument list
        short.append(n.replace("ol::", ""))
    return short


def compile_one(src, defs, keep_asm=None):
    with tempfile.TemporaryDirectory() as d:
        cmd = ["/opt/rocm/bin/hipcc", *FLAGS, *defs, "-c", os.path.join(CSRC, src), "-o",
               os.path.join(d, "x.o")]
        p = subprocess.run(cmd, capture_output=True, text=True)
        if p.returncode:
            sys.stderr.write(p.stderr)
            raise SystemExit(f"compile of {src} failed")
        if keep_asm:
            cmd = ["/opt/rocm/bin/hipcc", *[f for f in FLAGS if not f.startswith("-Rpass")], *defs,
                   "-S", os.path.join(CSRC, src), "-o", os.path.join(keep_asm, src + ".s")]
            subprocess.run(cmd, check=True, capture_output=True)
        return p.stderr


def parse(text):
    rows, cur = [], None
    for line in text.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\S+) \[-Rpass", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = m.group(2)
    return rows


def main(argv):
    defs = [a for a in argv if a.startswith("-D")]
    out = None
    flt = None
    asm = None
    it = iter(argv)
    for a in it:
        if a == "--out":
            out = next(it)
        elif a == "--filter":
            flt = next(it)
        elif a == "--asm":
            asm = next(it)
            os.makedirs(asm, exist_ok=True)
    with ThreadPoolExecutor(max_workers=len(SOURCES)) as pool:
        texts = list(pool.map(lambda s: compile_one(s, defs, asm), SOURCES))
    rows = [r for t in texts for r in parse(t)]
    names = demangle([r["name"] for r in rows])
    lines = [f"# static kernel resources, gfx950, flags: {' '.join(defs) or '(product build)'}",
             f"# {'kernel':<64} {'VGPR':>5} {'SGPR':>5} {'sSpill':>6} {'vSpill':>6} {'scratch':>7} "
             f"{'waves':>5} {'LDS':>5}"]
    for r, n in sorted(zip(rows, names), key=lambda x: x[1]):
        if flt and flt not in n:
            continue
        lines.append(f"  {n:<64} {r.get('VGPRs', '?'):>5} {r.get('TotalSGPRs', '?'):>5} "
                     f"{r.get('SGPRs Spill', '?'):>6} {r.get('VGPRs Spill', '?'):>6} "
                     f"{r.get('ScratchSize [bytes/lane]', '?'):>7} "
                     f"{r.get('Occupancy [waves/SIMD]', '?'):>5} "
                     f"{r.get('LDS Size [bytes/block]', '?'):>5}")
    text = "\n".join(lines) + "\n"
    sys.stdout.write(text)
    if out:
        with open(out, "w") as f:
            f.write(text)


if __name__ == "__main__":
    main(sys.argv[1:])
