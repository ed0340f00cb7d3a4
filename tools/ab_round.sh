#!/bin/bash
# Developer tool (GPU box): the round's kept changes against their round-3 forms in ONE call (same device): A = the round-3 attention /
# range-probe kernels (cdfo_amd/lib/r03like, built from git ref fa086e1 for those two files) with the round-4 schedule switches off,
# B = the current build.  usage: tools/ab_round.sh [rounds]
N="${1:-3}"
for i in $(seq 1 $N); do
  for E in "CDFO_LIB_PATH=$PWD/cdfo_amd/lib/r03like/libcdfo_hip.so CDFO_ATTN_PV3=1 CDFO_UDSA_STREAM=0" "CDFO_ATTN_PV3=0"; do
    env $E python bench.py --steps 12 --warmup 3 --no-parity --no-extra-modes --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$(echo $E | sed 's#/[^ ]*/cdfo_amd#cdfo_amd#')', r['value'], r['ms_per_step'], r['ranks'][0]['device']['pci'])"
  done
done
