#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
python -m pytest tests -m gpu -q -x 2>&1 | tail -12
for mode in last spot record; do
  echo "== $mode f32"
  timeout 300 python bench.py --mode $mode --no-cpu-baseline --steps 30 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
done
echo "== cooke spot"; timeout 300 python bench.py --mode spot --workload cooke --no-cpu-baseline --steps 30 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
echo "== record forced vector"; OL_TRACE_RPT=2 timeout 300 python bench.py --mode record --no-cpu-baseline --steps 30 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
