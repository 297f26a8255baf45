# rocprofv3 kernel statistics of the three non-headline configurations
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_cfg
mkdir -p $O
cd $R
for cfg in "resnet50_bilstm 16 64" "twostream_transformer 8 128" "i3d 16 64"; do
  set -- $cfg
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$1 -- python3 tools/run_config.py $1 $2 $3 5 > $O/$1.log 2>&1
  tail -1 $O/$1.log
done
find $O -name "*kernel_stats.csv" | head
