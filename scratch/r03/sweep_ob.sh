#!/bin/bash
# outer-block width of the two-level update on the three systems (21-point stencil, 5-point config 2, 100^3 cube)
run() { python bench.py --quick --steps $3 --warmup 2 --workload $1 $2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1 $TAG', 'step', round(d['ms_per_step'],3), 'factor', round(d['ms_factor'],3), 'solve', round(d['ms_solve'],3), 'res %.1e' % d['rel_residual'])"; }
for cfg in "base::" "ob128:128:0" "ob128m500:128:500" "ob128m1000:128:1000" "ob256m1000:256:1000" "ob128m2000:128:2000"; do
  IFS=: read TAG OB M <<< "$cfg"
  if [ -n "$OB" ]; then export KVX_OUTER_BLOCK=$OB KVX_TWO_LEVEL_M=$M; else unset KVX_OUTER_BLOCK KVX_TWO_LEVEL_M; fi
  export TAG
  run stencil21 "" 6
  run lap2d "" 20
done
unset KVX_OUTER_BLOCK KVX_TWO_LEVEL_M
for cfg in "base::" "ob128in1024::"; do
  IFS=: read TAG OB M <<< "$cfg"; export TAG
  run lap3d "" 3
done
