#!/bin/bash
run() { python bench.py --no-cpu-baseline --no-ipm --steps 30 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3), round(d['ms_factor'],3), round(d['ms_solve'],3), '%.1e' % d['rel_residual'])"; }
run default
KVX_TWO_LEVEL_M=1200 KVX_OUTER_BLOCK=1024 run "tl1200/ob1024"
KVX_TWO_LEVEL_M=1800 KVX_OUTER_BLOCK=1024 run "tl1800/ob1024"
KVX_TWO_LEVEL_M=1200 KVX_OUTER_BLOCK=2048 run "tl1200/ob2048"
KVX_TWO_LEVEL_M=600 KVX_OUTER_BLOCK=1024 run "tl600/ob1024"
run default
