cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_i3d
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 tools/run_config.py i3d 4 64 5 > $O/run.log 2>&1
tail -n 1 $O/run.log
cut -c1-140 $(find $O -name "*kernel_stats.csv" | head -1) | head -8
