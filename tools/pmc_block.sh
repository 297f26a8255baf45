# diagnostic: issue / wait / LDS / MFMA counters of the fused-block timing run (two passes of 8 SQ counters)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/pmc_block
rm -rf $O; mkdir -p $O
cd $R
timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $O/a -- python3 tools/time_block.py 640 > $O/a.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/b -- python3 tools/time_block.py 640 > $O/b.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, collections, os
root = os.environ.get('GRAFT_REPO_ROOT', os.getcwd()) + '/gpurun_out/pmc_block'
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for f in glob.glob(root + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][-40:]
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[k][r['Counter_Name']] += 1
for k, v in acc.items():
    if 'asl' in k:
        print(k, {c: round(x / cnt[k][c] / 1e6, 2) for c, x in sorted(v.items())})
PY
