cd /root/repo
python -m pytest tests/test_gpu_forward.py tests/test_gpu_backward.py -x -q -m gpu 2>&1 | tail -30
