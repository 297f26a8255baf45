# kernel trace of one ResNet-50 + BiLSTM forward on a single stream (B=4, T=64: one 256-frame chunk), per launch
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_r50
rm -rf $O; mkdir -p $O
cd $R
export ASL_STREAMS=1
timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 tools/run_config.py resnet50_bilstm 4 64 3 > $O/run.log 2>&1
tail -n 1 $O/run.log
