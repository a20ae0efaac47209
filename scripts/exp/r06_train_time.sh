R=$GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest $R/tests/test_gpu_train.py -x -q -m gpu 2>&1 | tail -3
python3 $R/scripts/time_train_step.py 2>&1 | grep -v amdgpu.ids | tail -12
