#!/bin/bash
run() { python bench.py --quick --steps $3 --warmup 2 --workload $1 $2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1 $TAG', 'step', round(d['ms_per_step'],3), 'factor', round(d['ms_factor'],3), 'solve', round(d['ms_solve'],3), 'res %.1e' % d['rel_residual'])"; }
for D in 0 64 1000 100000000; do
  export KVX_SYRK_DEEP_TILES=$D TAG=deep$D
  run stencil21 "" 6
  run lap3d "" 3
done
