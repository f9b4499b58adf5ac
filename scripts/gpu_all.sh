cd /root/repo
timeout 300 python -m pytest tests/test_gpu_forward.py tests/test_gpu_backward.py -x -q -m gpu 2>&1 | tail -8
timeout 300 python bench.py --steps ${1:-3} --warmup ${2:-1} --no-cpu-baseline 2>&1 | tail -2 | cut -c1-1500
