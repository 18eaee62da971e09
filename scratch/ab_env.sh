#!/bin/bash
# usage: ab_env.sh "<label>=<ENV=val ENV=val>" ...   (config 2 and the 1400 grid, bench --quick)
mkdir -p gpurun_out/s2
for spec in "$@"; do
  label=${spec%%=*}; envs=${spec#*=}
  for g in 1000 1400; do
    env $envs timeout -k 5 120 python bench.py --quick --steps 30 --warmup 8 --grid $g 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-28s g=$g step %.3f factor %.3f solve %.3f'%('$label',d['ms_per_step'],d['ms_factor'],d['ms_solve']))"
  done
done 2>&1 | tee -a gpurun_out/s2/ab_env.log
