#!/bin/bash
# GPU box: record-block plane stride A/B (OPTILAND_RECORD_ALIGN, bytes; the stride is n rounded
# up to a multiple of it): 2 MiB alignment (product) against SKEWED strides, where consecutive
# planes do not start on the same 2 MiB phase.  Interleaved, one box.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
OUT=$R/gpurun_out/ab_stride.txt; mkdir -p $R/gpurun_out; : > $OUT
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('   kernel_ms=%.4f moved=%.0f GB/s frac=%.3f'%(r['kernel_ms'],r['achieved'],r['frac']))"; }
echo "# $(date -u) double Gauss 1e7 rays record-all, 30 launches per arm, align in bytes ('' = product: 2 MiB)" >> $OUT
for dt in f32 f64; do
for rep in 1 2 3 4; do
  aligns="product 2097408 2101248 2105344 2162688 2228224 3145728 4198400 524288 65536"
  [ $((rep % 2)) -eq 0 ] && aligns=$(echo $aligns | tr ' ' '\n' | tac | tr '\n' ' ')   # alternate the order
  for a in $aligns; do a=${a/product/}
    echo -n "$dt align=${a:-product}" >> $OUT
    OPTILAND_RECORD_ALIGN=$a python bench.py --dtype $dt --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | show >> $OUT
  done
done
done
cat $OUT
