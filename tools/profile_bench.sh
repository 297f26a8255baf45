# The four rocprofv3 passes behind profiles/<tag>_*: kernel statistics, FETCH_SIZE, WRITE_SIZE, SQ + GRBM counters,
# all of the default bench.py command.  bash tools/profile_bench.sh <tag>   (on the GPU box)
set -e
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
rm -rf $O
mkdir -p $O
cd $R
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_stats.log 2>&1
echo stats done
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_fetch.log 2>&1
echo fetch done
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_write.log 2>&1
echo write done
timeout -s KILL 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc_sq -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_sq.log 2>&1
echo sq done
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_grbm -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_grbm.log 2>&1 || echo "GRBM pass failed"
echo grbm done
CH=$(grep '^{' $O/bench_stats.log | tail -1 | python3 -c "import json,sys;print(json.loads(sys.stdin.read())['config']['frames_per_chunk'])")
python3 tools/pmc_summary.py --json $O/${TAG}_counters.json --frames-per-chunk $CH $(find $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_grbm -name "*counter_collection.csv") > $O/${TAG}_pmc_summary.csv
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/${TAG}_kernel_stats.csv
timeout -k 10 200 python3 bench.py > $O/bench_final.log 2>&1
tail -1 $O/bench_final.log
ls $O
