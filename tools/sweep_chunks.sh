# frames per chunk x streams x clips per step: one bench line each (clips/s, ms per step)
set -e
O=gpurun_out/sweep
mkdir -p $O
for cfg in "20 640 2" "40 640 2" "40 1280 2" "20 1280 1" "40 1280 1" "30 640 3" "60 1920 2" "20 640 2"; do
  set -- $cfg
  out=$(timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 3 --clips $1 --chunk $2 --streams $3 2>/dev/null | tail -1) || exit 1
  echo "clips=$1 chunk=$2 streams=$3 $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["serial_frac"])')" | tee -a $O/sweep.log
done
