#!/bin/bash
# Same-box A/B of bench.py under two environments, interleaved rounds: tools/ab_bench.sh "<env A>" "<env B>" [rounds]
# (each run: python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-infer --no-kernel-events; prints ms_per_step)
a="$1"; b="$2"; n=${3:-2}
for r in $(seq 1 $n); do
  for e in "$a" "$b"; do
    out=$(env $e python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-infer --no-kernel-events 2>/dev/null | tail -1)
    echo "[$e] $(echo "$out" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms", d["value"], "img/s")')"
  done
done
