set -x
cd $GRAFT_REPO_ROOT
python -c "import torch;print(torch.cuda.is_available(), torch.cuda.get_device_name(0))"
make -s -C oracle
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -40
