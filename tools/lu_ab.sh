#!/bin/bash
# bash tools/lu_ab.sh : the KLU cases of bench_extra.py with the round-3 kernel of the LDS fronts (KVX_LU_WP=0) and the
# one-wavefront-panel kernel, in alternating runs on one box (run through gpurun from the repo root)
for rep in 1 2; do
for wp in 0 1; do
  KVX_LU_WP=$wp python3 bench_extra.py --cases klu3,lu2d 2>/dev/null | python3 -c "
import json,sys
for line in sys.stdin:
    d=json.loads(line); print('wp=$wp', d['case'][:12], 'refactor', round(d['ms_refactor_dev'],3), 'solve', round(d.get('ms_solve_dev', d.get('ms_solve_dev_incl_upload',0)),3), 'res', d.get('residual_inf', d.get('rel_residual')))"
done
done
