#!/bin/bash
# A/B of two builds of the library on the same box: bench.py headline, alternating
for rep in 1 2 3; do
  for v in v0 v5; do
    KVX_LIB_PATH=$GRAFT_REPO_ROOT/scratch/libkvxhip_$v.so python bench.py --no-cpu-baseline --no-ipm --steps 30 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],3), round(d['ms_factor'],3), round(d['ms_solve'],3))"
  done
done
