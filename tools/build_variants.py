#!/usr/bin/env python
"""Build A/B variants of the HIP library with other compile-time knobs (build container;
hipcc cross-compiles).  Output: optiland_amd/lib/variant_<name>.so -- git-ignored like the
product library, travels to the GPU box, selected there through OPTILAND_HIP_LIBRARY
(tools/gpu_ab_variants.sh).  The PRODUCT library is never touched.

    python tools/build_variants.py [name ...]      # default: all
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from optiland_amd import build as B  # noqa: E402

VARIANTS = {
    # round 6 (names start with "o": the ones .gpurunignore lets travel): the three experiments
    # on the one-ray-per-lane Newton / polarised kernels (trace_kernel.hip: OL_RECORD_DIRECT,
    # OL_PRT_SCALAR_BASE, OL_RECORD_ARGS_FRESH) and the packed PRT updates, against the product
    # (rows through a copy, per-lane PRT bases, fresh record arguments, packed PRT)
    # the polarised Zernike fp32 pair (OL_POLZ_PAIR): occupancy requests (108 VGPRs = 4 waves by
    # itself; 5 waves: 11 VGPRs to scratch, 6: 39), and the pair on two SCALAR rays per lane
    "o6_pair_off": ["-DOL_POLZ_PAIR=0"],
    "o6_pair_w5": ["-DOL_POLZ_PAIR_WAVES=5"],
    "o6_pair_w6": ["-DOL_POLZ_PAIR_WAVES=6"],
    "o6_pair_unpacked": ["-DOL_POLZ_PACKED=0"],
    "o6_prt_unpacked": ["-DOL_PRT_PACKED=0"],
    "o6_rows_direct": ["-DOL_RECORD_DIRECT=1"],
    "o6_prt_scalar_base": ["-DOL_PRT_SCALAR_BASE=1"],
    "o6_rec_args_held": ["-DOL_RECORD_ARGS_FRESH=0"],
    "o6_r05": ["-DOL_PRT_PACKED=0", "-DOL_RECORD_ARGS_FRESH=0"],
    # BASELINE.json north_star: "surface coefficients staged in LDS" 
    "lds_table": ["-DOL_TABLE_IN_LDS=1"],
    # store flavour: plain stores for the one-ray-per-lane layout,
    # non-temporal for the 16-byte vector layout
    "plain_stores": ["-DOL_NT_SCALAR=0"],
    "nt_vector": ["-DOL_NT_VECTOR=1"],
    # workgroup size
    "block128": ["-DOL_TRACE_BLOCK=128"],
    "block512": ["-DOL_TRACE_BLOCK=512"],
    # register-allocator occupancy request for the fp32 polarised Newton kernel (C5)
    "polnr_waves0": ["-DOL_POLNR_WAVES=0"],
    "polnr_waves7": ["-DOL_POLNR_WAVES=7"],
    "polnr_waves8": ["-DOL_POLNR_WAVES=8"],
    # plane accesses addressed per lane with 64-bit VGPR addresses (round-1 form) instead of
    # SGPR base + one shared 32-bit lane offset
    "vaddr": ["-DOL_SADDR=0"],
    # low-order Zernike as one polynomial: no load fence between the sag
    # and the gradient chains / no unrolled instances (every degree runs the loops).  The
    # level form itself needs no build: OPTILAND_HIP_ZERNIKE_MONO=0 at ol_system_create.
    "zmono_nosplit": ["-DOL_ZERN_MONO_SPLIT=0"],
    "zmono_loops": ["-DOL_ZERN_MONO_FIXED=0"],
    # hot-block prefetch policy (trace_kernel.hip): not yet measured for the fp64 kernels
    # (32 more live SGPRs; static SGPR-spill counts drop without it) and for the fused
    # spot / OPD Newton kernels -- tools/gpu_ab_prefetch.sh
    "noprefetch_f64": ["-DOL_PREFETCH_F64=0"],
    "noprefetch_fused_nr": ["-DOL_FUSED_NR_PREFETCH=0"],
    "nr_prefetch": ["-DOL_NR_PREFETCH=1"],
    # round 3: table rows / kernel arguments re-read where they are used (SurfFetched,
    # kernargs()) instead of held in SGPRs; windowed coefficient stream of the unrolled
    # Zernike polynomial.  (The round-2 library itself is built from its commit as
    # variant_r02.so by tools/build_r02_variant.sh.)
    # fetch levels per kernel class (trace_kernel.hip: OL_FETCH_*)
    "nr32_0": ["-DOL_FETCH_NR_F32=0"],
    "nr32_1": ["-DOL_FETCH_NR_F32=1"],
    "nr32_2": ["-DOL_FETCH_NR_F32=2"],
    "nr64_0": ["-DOL_FETCH_NR_F64=0"],
    "nr64_1": ["-DOL_FETCH_NR_F64=1"],
    "lean64_0": ["-DOL_FETCH_LEAN_F64=0"],
    "lean64_2": ["-DOL_FETCH_LEAN_F64=2"],
    "lean32_1": ["-DOL_FETCH_LEAN_F32=1"],
    "lean32_2": ["-DOL_FETCH_LEAN_F32=2"],
    "nr32_2_waves0": ["-DOL_FETCH_NR_F32=2", "-DOL_POLNR_WAVES=0"],
    "nr32_0_waves0": ["-DOL_FETCH_NR_F32=0", "-DOL_POLNR_WAVES=0"],
    "zmono_ahead2": ["-DOL_ZERN_MONO_AHEAD=2"],
    "zmono_chunk16": ["-DOL_ZERN_MONO_CHUNK=16"],
    "polnr_waves6": ["-DOL_POLNR_WAVES=6"],
    # fp64 division / square root: IEEE library sequences instead of the hardware-seed +
    # two refinement steps form (surface_math.h: OL_FAST_F64)
    "fast64_0": ["-DOL_FAST_F64=0"],
    # fp32 unrolled Zernike polynomial through the 8-dword window (as fp64) instead of as two
    # blocks (sag, then gradient)
    "zmono_window32": ["-DOL_ZERN_MONO_F32_TWO_BLOCKS=0"],
    # round 4: IEEE square root + quotients in the ray generator and the reference-sphere
    # intersection (the round-3 form) instead of the hardware reciprocal square root / seeds
    "raygen_ieee": ["-DOL_RAYGEN_RSQ=0", "-DOL_WAVEFRONT_FAST=0"],
    # fp64 quotient with two Newton steps on the reciprocal (rounds 3 - 4a) instead of one
    "div64_2steps": ["-DOL_DIV_F64_STEPS=2"],
    # round 5: the two quotients of the conic intersection from two reciprocal seeds (rounds
    # 3-4) instead of one shared; the aligned normal formed explicitly (rounds 1-4) instead of
    # the sign carried on Snell's root; both = the arithmetic of round 4's library
    "div2_off": ["-DOL_DIV2_F64=0"],
    "snell_r04": ["-DOL_SNELL_SIGN_ON_ROOT=0"],
    "arith_r04": ["-DOL_DIV2_F64=0", "-DOL_SNELL_SIGN_ON_ROOT=0"],
    # the opt-in reference-formula conic root compiled out (what its launch-uniform branch costs
    # the kernels that never take it)
    "no_refroot": ["-DOL_REFERENCE_ROOT=0"],
    # occupancy request for the lean fp64 kernels (the generating record-all form: 82 VGPRs = 5
    # waves by itself; 79 / 6 waves on request without spills; 72 / 7 waves with 12 B of scratch)
    "lean64_w6": ["-DOL_LEAN_F64_WAVES=6"],
    "lean64_w7": ["-DOL_LEAN_F64_WAVES=7"],
    # the fused OPD kernel with one ray per lane everywhere (round 5 default: two without a
    # Newton surface, launches of >= 2^20 rays)
    "opd_one": ["-DOL_OPD_TWO_RAYS=0"],
    # Newton stop rule of rounds 1-4: a ray whose residual stops halving leaves at ANY level
    # (round 5: only at the rounding floor of sag - z)
    "old_stall": ["-DOL_NR_STALL_ULPS=0"],
    # without the end-of-trace test behind OL_STATUS_NAN_DIRECTION (two compares per ray)
    "o_nobit": ["-DOL_NAN_DIRECTION_BIT=0"],
}


def build(name):
    flags = ["--offload-arch=" + B.ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on",
             "-fno-math-errno"] + VARIANTS[name]
    out = os.path.join(B.LIBDIR, f"variant_{name}.so")
    objdir = os.path.join(B.LIBDIR, f"variant_{name}_obj")
    os.makedirs(objdir, exist_ok=True)
    objs, jobs = [], []
    for src in B.SOURCES:
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        jobs.append([B._hipcc(), *flags, "-c", os.path.join(B.CSRC, src), "-o", o])
        objs.append(o)
    with ThreadPoolExecutor(max_workers=4) as pool:
        list(pool.map(subprocess.check_call, jobs))
    subprocess.check_call([B._hipcc(), "--offload-arch=" + B.ARCH, "-shared", "-fPIC", *objs,
                           "-o", out])
    for o in objs:
        os.remove(o)
    os.rmdir(objdir)
    return out


if __name__ == "__main__":
    for nm in (sys.argv[1:] or list(VARIANTS)):
        print(build(nm))
