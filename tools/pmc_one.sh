#!/bin/bash
# SQ counters of one kind of launch of a plan, alone on the GPU: bash tools/pmc_one.sh <spec> <frames> <OP_KIND> <kernel name fragment> [tag]
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${5:-pmc_one}
rm -rf $O && mkdir -p $O
cd $R
export ONLY=$3
timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/sq -- python3 tools/time_plan_ops.py $1 $2 3 > $O/sq.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $O/sq2 -- python3 tools/time_plan_ops.py $1 $2 3 > $O/sq2.log 2>&1 || echo "second pass failed"
python3 - $4 $(find $O -name "*counter_collection.csv") <<'PY'
import csv, sys, collections
frag = sys.argv[1]
acc = collections.defaultdict(list)
for f in sys.argv[2:]:
    for r in csv.DictReader(open(f)):
        if frag in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:28s} launches {len(v):3d}  per launch {sum(v)/len(v):16.1f}")
PY
