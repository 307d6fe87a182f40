#!/bin/bash
# A/B on ONE box (box-to-box variance is a few %): tools/ab.sh "ENV_A=.." "ENV_B=.." [rounds] [bench args...]
A="$1"; B="$2"; N=${3:-4}; shift 3
for i in $(seq $N); do
  for e in "$A" "$B"; do
    v=$(env $e python bench.py --no-cpu-baseline --no-profile "$@" 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.readline()); print(j['median_ms_per_step'])")
    echo "$e $v"
  done
done | sort | awk '{s[$1]=s[$1]" "$2; n[$1]++; t[$1]+=$2} END {for (k in s) printf "%-40s mean %.4f  :%s\n", k, t[k]/n[k], s[k]}'
