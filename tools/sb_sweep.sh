#!/bin/bash
# Stereo R-CNN shape (600x1987) step kernel: workgroup size x sub-tiles per workgroup, through bench.py --workload srcnn.
# Variant libraries are built with -DADV_SHIFT_BLOCK / -DADV_SHIFT_UNROLL into tools/_build/ (see DESIGN.md).
R=${GRAFT_REPO_ROOT:-$(pwd)}
for lib in "" $R/tools/_build/libadv_sb*_u*.so; do
  for mode in "" "--alternate"; do
    echo "== ${lib:-default(256,1)} $mode"
    ADVENGINE_LIB=$lib python3 $R/bench.py --workload srcnn --pairs 64 --steps 3 --warmup 1 --no-cpu-baseline $mode 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('   pairs/s %.0f  kernel %.3f ms  frac %.3f' % (d['value'], r['avg_launch_ms'], r['frac']))"
  done
done
