#!/bin/bash
run() { python bench.py --quick --steps $3 --warmup 2 --workload $1 $2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1 $TAG', 'step', round(d['ms_per_step'],3), 'factor', round(d['ms_factor'],3), 'solve', round(d['ms_solve'],3), 'res %.1e' % d['rel_residual'])"; }
for PT in ${PTS:-1000000000 1 300 1500 5000 20000}; do
  export KVX_PAIR_TILES=$PT TAG=pair$PT
  run stencil21 "" 6
  run lap2d "" 20
  run lap3d "" 3
done
