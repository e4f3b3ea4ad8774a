#!/usr/bin/env python
"""Copy the judged rocprofv3 summaries from gpurun_out/ (scratch) into profiles/
(tracked) and derive profiles/traffic.json (HBM bytes per trace_kernel launch).

HBM bytes per launch = 2 * FETCH_SIZE + WRITE_SIZE (KiB -> bytes): on gfx950 this
rocprofv3 reports exactly half of the bytes of a wide coalesced read
(MI355X_MICROARCH.md, HBM section) -- confirmed here on torch's own copy kernels in
the same trace (40 MB read -> FETCH_SIZE 19.55 MiB; 40 MB written -> WRITE_SIZE
39062.5 KiB exactly).
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
PROF = os.path.join(ROOT, "profiles")
ROUND = sys.argv[1] if len(sys.argv) > 1 else "r01"
TAGS = {  # tag -> traffic key
    # record-mode bench runs use the zero-copy object row by default (":alias")
    "dg_f32": "double_gauss:f32:record:alias",
    "dg_f64": "double_gauss:f64:record:alias",
    "rc_f32": "rc_asphere:f32:record:alias",
    "zf_f32": "zernike_fresnel:f32:record:alias",
    "dg_f32_last": "double_gauss:f32:last",
    "dg_f32_copy": "double_gauss:f32:record",
    "dg_f32_spot": "double_gauss:f32:spot",
    "dg_f64_spot": "double_gauss:f64:spot",
    # round 3 (tools/gpu_round3.sh): generation fused into the record-all launch is the
    # default bench mode; the fp64 / Newton kernels whose table access changed
    "r03_dg_f32_gen": "double_gauss:f32:gen",
    "r03_zf_f32_gen": "zernike_fresnel:f32:gen",
    "r03_rc_f32_gen": "rc_asphere:f32:gen",
    "r03_dg_f64_gen": "double_gauss:f64:gen",
    "r03_zf_f64": "zernike_fresnel:f64:record:alias",
    "r03_rc_f64": "rc_asphere:f64:record:alias",
    "r03_z_opd": "zernike:f64:opd",
    # round 4 (tools/gpu_r04_final.sh): the default line with the placed record block
    "r04_dg_f32_gen": "double_gauss:f32:gen",
    "r04_dg_f64_gen": "double_gauss:f64:gen",
    "r04_rc_f32_gen": "rc_asphere:f32:gen",
    "r04_zf_f32_gen": "zernike_fresnel:f32:gen",
    # round 5 (tools/gpu_r05.sh prof): the final library
    "r05_dg_f32_gen": "double_gauss:f32:gen",
    "r05_dg_f64_gen": "double_gauss:f64:gen",
    "r05_rc_f32_gen": "rc_asphere:f32:gen",
    "r05_zf_f32_gen": "zernike_fresnel:f32:gen",
    # round 6 (tools/gpu_r06.sh prof): the final library
    "r06_dg_f32_gen": "double_gauss:f32:gen",
    "r06_dg_f64_gen": "double_gauss:f64:gen",
    "r06_rc_f32_gen": "rc_asphere:f32:gen",
    "r06_zf_f32_gen": "zernike_fresnel:f32:gen",
}


def counter_mean(tag, sub, ctr):
    g = glob.glob(os.path.join(OUT, f"prof_{tag}", sub, "**", "*counter_collection.csv"),
                  recursive=True)
    if not g:
        return None
    vals = []
    with open(g[0]) as fh:
        for r in csv.DictReader(fh):
            if r.get("Counter_Name") == ctr and "trace_kernel" in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]))
    return sum(vals) / len(vals) if vals else None


def main():
    os.makedirs(PROF, exist_ok=True)
    traffic = {}
    tpath = os.path.join(PROF, "traffic.json")
    if os.path.exists(tpath):
        traffic = json.load(open(tpath))
    for tag, key in TAGS.items():
        d = os.path.join(OUT, f"prof_{tag}")
        if not os.path.isdir(d):
            continue
        lines = [f"# {ROUND} {key}: rocprofv3 --kernel-trace --stats --output-format csv -- "
                 f"python bench.py --steps 20 --warmup 5 --settle 0 --no-cpu-baseline (+ workload flags)"]
        bench = [l for l in open(os.path.join(d, "stats.log")) if l.startswith("{")]
        if bench:
            b = json.loads(bench[-1])
            lines.append(f"# bench line under the profiler: value={b['value']:.4g} rs/s "
                         f"kernel_ms(HIP events)={b['roofline']['kernel_ms']:.4f} "
                         f"achieved={b['roofline']['achieved']:.0f} GB/s")
        g = glob.glob(os.path.join(d, "stats", "**", "*kernel_stats.csv"), recursive=True)
        if g:
            lines.append("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs")
            with open(g[0]) as fh:
                for r in list(csv.DictReader(fh))[:6]:
                    nm = r["Name"]
                    nm = nm if len(nm) < 110 else nm[:107] + "..."
                    lines.append(f"\"{nm}\",{r['Calls']},{r['TotalDurationNs']},"
                                 f"{float(r['AverageNs']):.1f},{r['Percentage']},{r['MinNs']},{r['MaxNs']}")
        f_kib = counter_mean(tag, "pmc_fetch", "FETCH_SIZE")
        w_kib = counter_mean(tag, "pmc_write", "WRITE_SIZE")
        if f_kib is not None and w_kib is not None:
            hbm = 2 * f_kib * 1024 + w_kib * 1024
            traffic[key] = hbm
            lines.append(f"# PMC (separate passes): FETCH_SIZE={f_kib:.1f} KiB (x2 gfx950 correction "
                         f"-> {2*f_kib*1024/1e6:.1f} MB), WRITE_SIZE={w_kib:.1f} KiB "
                         f"({w_kib*1024/1e6:.1f} MB) per trace_kernel launch; HBM bytes = {hbm/1e6:.1f} MB")
        name = tag[len(ROUND) + 1:] if tag.startswith(ROUND + "_") else tag
        open(os.path.join(PROF, f"{ROUND}_{name}_rocprof.txt"), "w").write("\n".join(lines) + "\n")
        print("\n".join(lines[:4]))
    box = os.path.join(OUT, f"{ROUND}_box.txt")
    if os.path.exists(box):
        traffic["_meta"] = {"round": ROUND,
                            "box": " ".join(open(box).read().split())[:200],
                            "note": "rows of earlier rounds were taken on those rounds' boxes"}
    json.dump(traffic, open(tpath, "w"), indent=1, sort_keys=True)
    bd = os.path.join(OUT, f"{ROUND}_bench_default.json")
    if os.path.exists(bd):
        open(os.path.join(PROF, f"{ROUND}_bench_default.json"), "w").write(open(bd).read())


if __name__ == "__main__":
    main()
