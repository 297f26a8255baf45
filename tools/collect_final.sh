set -e
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/fin
timeout -k 10 400 python tools/time_configs.py 20 > gpurun_out/fin/configs.jsonl 2> gpurun_out/fin/configs.err
echo configs done
timeout -k 10 100 python tools/time_plan_ops.py resnet18_mean 640 20 > gpurun_out/fin/ops_resnet18_640.log 2>&1
timeout -k 10 100 python tools/time_plan_ops.py resnet50_bilstm 512 20 > gpurun_out/fin/ops_resnet50_512.log 2>&1
timeout -k 10 100 python tools/time_i3d_ops.py 16 10 > gpurun_out/fin/ops_i3d_16.log 2>&1
echo ops done
bash tools/prof_configs.sh > gpurun_out/fin/prof_configs.log 2>&1
echo prof_configs done
