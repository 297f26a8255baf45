for cfg in "20 640 2" "20 1280 1" "20 427 3" "20 320 4" "20 320 2" "30 640 3" "30 960 2" "40 640 2" "40 1280 2" "10 640 1" "10 320 2"; do
  set -- $cfg
  echo -n "clips $1 chunk $2 streams $3: "
  timeout -k 10 120 python bench.py --clips $1 --chunk $2 --streams $3 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r['roofline']['frac'])" || exit 1
done
