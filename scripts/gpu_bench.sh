cd /root/repo
timeout 120 python __graft_entry__.py smoke 2>&1 | tail -3
timeout 300 python bench.py --steps ${1:-3} --warmup ${2:-1} --no-cpu-baseline 2>&1 | tail -5
