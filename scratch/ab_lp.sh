#!/bin/bash
one() { timeout -k 5 200 python bench_extra.py --cases lp4b 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$1', 'loop it/s %.1f whole %.1f'%(d['iterations_per_s_loop_only'], d['value']))"; }
one default; one default
KVX_CHAIN_PRIO=0 one noprio; KVX_CHAIN_PRIO=0 one noprio
KVX_NO_GRAPH=1 one nograph
python bench.py --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench.py ipm', d['ipm']['value'], d['ipm']['loop_s'])"
