#!/bin/bash
# A/B of library builds on one box, alternating: bash tools/ab.sh "base v1" [reps] [bench args]
VARS=${1:-"base v1"}; REPS=${2:-3}; shift; shift
for rep in $(seq $REPS); do
  for v in $VARS; do
    KVX_LIB_PATH=$GRAFT_REPO_ROOT/scratch/libkvxhip_$v.so python bench.py --quick --steps 30 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'step', round(d['ms_per_step'],3), 'factor', round(d['ms_factor'],3), 'solve', round(d['ms_solve'],3), 'res %.1e' % d['rel_residual'])"
  done
done
