#!/bin/bash
# HBM-side bytes per launch of the kernels whose name contains <fragment>, over any command (two --pmc passes):
#   bash tools/pmc_bytes.sh <tag> <kernel name fragment> <program> [args...]      (run on the GPU box)
# bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB, the gfx950 read-side correction the bench's counters use.
set -e
TAG=$1; FRAG=$2; shift 2
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
rm -rf $O && mkdir -p $O
cd $R
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -- "$@" > $O/f.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w -- "$@" > $O/w.log 2>&1
python3 - "$FRAG" $(find $O -name "*counter_collection.csv") <<'PY'
import csv, sys, collections
frag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sys.argv[2:]:
    for r in csv.DictReader(open(f)):
        if frag in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0][-48:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    fs = sum(v["FETCH_SIZE"]) / max(1, len(v["FETCH_SIZE"])); ws = sum(v["WRITE_SIZE"]) / max(1, len(v["WRITE_SIZE"]))
    print(f"{k:48s} launches {len(v['FETCH_SIZE']):4d}  read {2 * fs * 1024 / 1e6:9.1f} MB  write {ws * 1024 / 1e6:9.1f} MB  total {(2 * fs + ws) * 1024 / 1e6:9.1f} MB per launch")
PY
rm -rf $O/f $O/w
