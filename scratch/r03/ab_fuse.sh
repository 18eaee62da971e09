#!/bin/bash
for rep in 1 2 3; do
  for mode in "--separate-calls" ""; do
    for w in lap2d stencil21; do
      python bench.py --quick --steps 20 --workload $w $mode 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$w', '$mode' or 'fused', 'step', round(d['ms_per_step'],3), 'factor', round(d['ms_factor'],3), 'solve', round(d['ms_solve'],3), 'res %.1e' % d['rel_residual'])"
    done
  done
done
python bench.py --quick --steps 5 --workload lap3d --separate-calls 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('lap3d sep', round(d['ms_per_step'],3))"
python bench.py --quick --steps 5 --workload lap3d 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('lap3d fused', round(d['ms_per_step'],3))"
