#!/bin/bash
# which of this round's switches makes the 2-rank gloo test hang?  each attempt under its own timeout
t() { echo "== $1"; env $1 timeout -k 5 150 python -u -m pytest "tests/test_dist_gpu.py::test_sharded_factor_solve_matches_single_process[2]" -x -q 2>&1 | tail -3; }
t "KVX_PAIR_TILES=1000000000"
t "KVX_FWD_NARROW_WGS=0"
t "KVX_WAVE_MERGE=0"
t "KVX_DIST_NO_TRIM=1"
