#!/bin/bash
# look-ahead of the two-level blocking: thresholds x outer block widths on the flop-bound cases
mkdir -p gpurun_out/s2
run() { # label env...
  local label=$1; shift
  env "$@" timeout -k 5 300 python bench_extra.py --cases lap3d,chol21 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    line = line.strip()
    if not line.startswith('{'): continue
    d = json.loads(line)
    print('%-34s %-28s factor %.2f ms solve %.2f ms %.1f GF/s resid %.1e' % ('$label', d.get('case','?')[:28], d.get('ms_factor',0), d.get('ms_solve',0), d.get('value',0), d.get('rel_residual',0)))
"
}
{
run off KVX_LOOKAHEAD=0
run la_6144_1024 KVX_LOOKAHEAD=1
run la_3072_512 KVX_LOOKAHEAD=1 KVX_TWO_LEVEL_M=3072 KVX_OUTER_BLOCK=512
run la_2048_512 KVX_LOOKAHEAD=1 KVX_TWO_LEVEL_M=2048 KVX_OUTER_BLOCK=512
run la_2048_256 KVX_LOOKAHEAD=1 KVX_TWO_LEVEL_M=2048 KVX_OUTER_BLOCK=256
run la_1024_256 KVX_LOOKAHEAD=1 KVX_TWO_LEVEL_M=1024 KVX_OUTER_BLOCK=256
run la_4096_1024 KVX_LOOKAHEAD=1 KVX_TWO_LEVEL_M=4096 KVX_OUTER_BLOCK=1024
} 2>&1 | tee gpurun_out/s2/ab_la.log
