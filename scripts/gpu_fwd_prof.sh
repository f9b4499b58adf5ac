set -e
cd /root/repo
python -m pytest tests/test_gpu_forward.py -x -q -m gpu 2>&1 | tail -3
python scripts/fwd_time.py ${1:-shared_training} ${2:-64} ${3:-600}
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_fwd -o fwd -- python /root/repo/scripts/fwd_time.py ${1:-shared_training} ${2:-64} ${3:-600} > /dev/null 2>&1
python /root/repo/scripts/prof_summary.py /root/repo/ gpurun_out/prof_fwd/fwd_results.db 12
