set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import torch;print(torch.cuda.is_available(), torch.cuda.get_device_name(0))"
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1 && tail -5 gpurun_out/pytest_gpu.log && bash scripts/gpu_final.sh
