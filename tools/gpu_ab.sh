#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
python tools/ab_bench.py --workload double_gauss --dtype f32 --mode record
python tools/ab_bench.py --workload double_gauss --dtype f64 --mode record
python tools/ab_bench.py --workload rc_asphere --dtype f32 --mode record
python tools/ab_bench.py --workload rc_asphere --dtype f32 --mode last
python tools/ab_bench.py --workload zernike_fresnel --dtype f32 --mode record
python tools/ab_bench.py --workload zernike_fresnel --dtype f32 --mode last
python tools/ab_bench.py --workload zernike_fresnel --dtype f64 --mode record
python tools/ab_bench.py --system-json tests/golden/aspheric_singlet.json --dtype f32 --mode last --tol 1e-12
python tools/ab_bench.py --system-json tests/golden/aspheric_singlet.json --dtype f64 --mode last --tol 1e-13
python tools/ab_bench.py --system-json tests/golden/nr_family.json --hy 1.0 --dtype f32 --mode last
