# rocprofv3 kernel statistics of one configuration: bash tools/prof_one.sh <spec> <B> <T> [tag]
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${4:-prof_one}
mkdir -p $O
cd $R
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw -- python3 tools/run_config.py $1 $2 $3 5 > $O/run.log 2>&1
cp $(find $O/raw -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf $O/raw
tail -1 $O/run.log
python3 - $O/kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:12]:
    print(f'{r["Name"][:80]:80s} calls {r["Calls"]:>5s} total {float(r["TotalDurationNs"])/1e6:8.2f} ms {100*float(r["TotalDurationNs"])/tot:5.1f}%  avg {float(r["AverageNs"])/1e3:8.1f} us')
PY
