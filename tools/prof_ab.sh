# Per-kernel times of the bench step on ONE stream (kernels do not overlap, so average durations are each kernel's own):
# bash tools/prof_ab.sh <tag> [env assignments...]   (on the GPU box; writes gpurun_out/<tag>/kernel_stats.csv)
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for kv in "$@"; do export "$kv"; done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 4 --warmup 2 --streams 1 --no-cpu-baseline > $O/bench_stats.log 2>&1
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf $O/stats
cut -d, -f1-4 $O/kernel_stats.csv | head -16
