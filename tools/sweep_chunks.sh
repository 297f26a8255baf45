#!/bin/bash
# Times bench.py over (frames per chunk, streams); prints clips/s per configuration.
for cfg in "256 2" "128 2" "128 4" "192 3" "64 4" "64 8" "128 3" "256 3" "512 1"; do
  set -- $cfg
  out=$(timeout -k 10 200 python bench.py --no-cpu-baseline --steps 30 --warmup 5 --chunk $1 --streams $2 2>/dev/null | tail -1) || exit 1
  echo "chunk $1 streams $2: $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
done
