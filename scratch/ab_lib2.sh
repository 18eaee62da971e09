#!/bin/bash
for rep in 1 2 3 4; do
  for v in head new; do
    KVX_LIB_PATH=$GRAFT_REPO_ROOT/scratch/libkvxhip_$v.so python bench.py --no-cpu-baseline --no-ipm --steps 30 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],3), round(d['ms_factor'],3), round(d['ms_solve'],3))"
  done
done
