#!/bin/bash
# same-box A/B of bench.py under environment switches: ab_env.sh "<bench args>" "VAR=val ..." "VAR=val ..." ...
# (each quoted group is one arm; "-" = no switches); two passes, prints samples/s and ms per arm
ARGS="$1"; shift
mkdir -p gpurun_out
for pass in 1 2; do
  i=0
  for arm in "$@"; do
    i=$((i+1))
    if [ "$arm" = "-" ]; then envs=""; else envs="$arm"; fi
    env $envs python bench.py --steps ${STEPS:-30} --warmup 10 --no_cpu_baseline $ARGS > gpurun_out/abenv_${i}_$pass.json 2> gpurun_out/abenv_${i}_$pass.err || tail -3 gpurun_out/abenv_${i}_$pass.err
    python - "$arm" gpurun_out/abenv_${i}_$pass.json <<'EOF2'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    print('%-50s %9.2f samples/s  %7.3f ms' % (sys.argv[1], d['value'], d['ms_per_step']))
except Exception as e:
    print(sys.argv[1], 'ERR', e)
EOF2
  done
done
