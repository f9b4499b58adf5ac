cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_infer -o infer -- python /root/repo/scripts/prof_inference.py > /root/repo/gpurun_out/infer_prof.log 2>&1
tail -2 /root/repo/gpurun_out/infer_prof.log
python /root/repo/scripts/geom_summary.py /root/repo/gpurun_out/prof_infer/infer_results.db "" 22
