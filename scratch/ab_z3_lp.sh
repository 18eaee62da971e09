#!/bin/bash
for z3 in 0.05 0.075 0.05 0.075; do
  export KVX_RELAX_Z3=$z3
  timeout -k 5 400 python bench_extra.py --cases lp4b,lp4c,lu2d,lp4a 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l)
    if 'iterations_per_s_loop_only' in d: print('z3=$z3', d['case'][:12], 'loop it/s %.1f'%d['iterations_per_s_loop_only'], 'factor %.3f solve %.3f'%(d.get('ms_kkt_factor',0),d.get('ms_kkt_solve',0)))
    else: print('z3=$z3', d['case'][:12], 'refactor %.2f ms solve %.2f'%(d['ms_refactor_dev'], d.get('ms_solve_dev_incl_upload',0)))"
done
