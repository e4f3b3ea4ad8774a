#!/usr/bin/env python3
"""Compile a FEW kernel instantiations of trace_kernel.hip by themselves (seconds instead of
the minutes of the full translation units) and report registers, spills and where the lane
(spill) instructions sit.  CPU only.

usage: kernel_probe.py [-DNAME=VALUE ...] [--keep DIR] 'trace_kernel<float,1,true,1,3,false>' ...
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "optiland_amd", "csrc")
sys.path.insert(0, os.path.join(ROOT, "tools"))
import asm_stats  # noqa: E402
import kernel_resources as kr  # noqa: E402

ARGS = {"trace_kernel": "TraceArgs", "spot_trace_kernel": "SpotArgs", "opd_trace_kernel": "OpdArgs"}


def main(argv):
    defs = [a for a in argv if a.startswith("-D")]
    keep = None
    specs = []
    it = iter(argv)
    for a in it:
        if a == "--keep":
            keep = next(it)
        elif not a.startswith("-D"):
            specs.append(a)
    src = ['#define OL_TRACE_TU 3', f'#include "{CSRC}/trace_kernel.hip"', "namespace ol {"]
    for sp in specs:
        name, targs = re.match(r"(\w+)<(.*)>", sp.replace(" ", "")).groups()
        T = targs.split(",")[0]
        src.append(f"template __global__ void {name}<{targs}>(const DevSurfHot<{T}>*, "
                   f"const DevSurfCold<{T}>*, const DevOptics<{T}>*, const {T}*, {ARGS[name]}<{T}>);")
    src.append("}")
    d = keep or tempfile.mkdtemp()
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, "probe.hip")
    with open(path, "w") as f:
        f.write("\n".join(src) + "\n")
    base = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
            "-ffp-contract=on", "-fno-math-errno", "--cuda-device-only", f"-I{CSRC}", *defs]
    p = subprocess.run(base + ["-Rpass-analysis=kernel-resource-usage", "-S", path, "-o",
                               os.path.join(d, "probe.s")], capture_output=True, text=True)
    if p.returncode:
        sys.stderr.write(p.stderr)
        raise SystemExit(1)
    rows = kr.parse(p.stderr)
    names = kr.demangle([r["name"] for r in rows])
    fns = asm_stats.functions(os.path.join(d, "probe.s"))
    for r, n in zip(rows, names):
        print(f"{n}: VGPR {r.get('VGPRs')} SGPR {r.get('TotalSGPRs')} sSpill {r.get('SGPRs Spill')} "
              f"vSpill {r.get('VGPRs Spill')} scratch {r.get('ScratchSize [bytes/lane]')} "
              f"waves {r.get('Occupancy [waves/SIMD]')}")
        if r["name"] in fns:
            tot, _ = asm_stats.analyse(fns[r["name"]])
            valu = sum(t["valu"] for t in tot.values())
            lane = sum(t["lane"] for t in tot.values())
            smem = sum(t["smem"] for t in tot.values())
            salu = sum(t["salu"] for t in tot.values())
            inloop = sum(t["lane"] for dep, t in tot.items() if dep > 0)
            print(f"    static: valu {valu} (lane {lane}, in loops {inloop})  smem {smem}  salu {salu}")
    if keep:
        print(f"kept {d}/probe.s")


if __name__ == "__main__":
    main(sys.argv[1:])
