#!/bin/bash
# Developer tool (GPU box): same-box A/B of bench.py under two environments.  usage: tools/ab_bench.sh "ENV_A=.." "ENV_B=.." [rounds]
A="$1"; B="$2"; N="${3:-2}"
for i in $(seq 1 $N); do
  for E in "$A" "$B"; do
    env $E python bench.py --steps 12 --warmup 3 --no-parity --no-extra-modes --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$E', r['value'], r['ms_per_step'])"
  done
done
