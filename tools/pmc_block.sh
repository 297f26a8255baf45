# diagnostic: LDS / MFMA counters of the fused-block timing run
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r2i/pmc_block
mkdir -p $O
cd $R
timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $O -- python3 tools/time_block.py 640 > $O/run.log 2>&1
python3 - <<'PY'
import csv, glob, collections, os
f = glob.glob(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r2i/pmc_block/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'][:60]
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'SQ_BUSY_CYCLES': cnt[k] += 1
for k, v in acc.items():
    if 'asl' in k:
        n = cnt[k]
        print(k, n, {c: round(x / n) for c, x in v.items()})
PY
