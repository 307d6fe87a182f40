#!/bin/bash
# Interleaved sweep on ONE box: tools/sweep.sh ROUNDS "ENV_A" "ENV_B" ... (each ENV is a space-free VAR=val or VAR=val,VAR2=val2)
N=$1; shift
for i in $(seq $N); do
  for e in "$@"; do
    v=$(env ${e//,/ } python bench.py --no-cpu-baseline --no-profile $BENCH_ARGS 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.readline()); print(j['median_ms_per_step'])")
    echo "$e $v"
  done
done | sort | awk '{s[$1]=s[$1]" "$2; n[$1]++; t[$1]+=$2} END {for (k in s) printf "%-50s mean %.4f  :%s\n", k, t[k]/n[k], s[k]}' | sort -k3
