#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_exch
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o ex -- \
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 $R/bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --force-exchange > $OUT/log.txt 2>&1
tail -1 $OUT/log.txt | cut -c1-200
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print(r['Name'][:70], r['Calls'], r['AverageNs'], r['Percentage'])
PY
