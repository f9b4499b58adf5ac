# round 5, call B: long inputs through the persistent decoder (tests + A/B against the per-step schedule), MFMA-utilisation PMC passes
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r05b; mkdir -p $O
( timeout 900 python -m pytest -q -x -m gpu tests/test_gpu_persist.py -s 2>&1 | grep -v "^$" | tail -30 ) > $O/tests.log 2>&1
tail -4 $O/tests.log
{
  echo "# decoder forward step, shared_training, batch 64, T = 300 (scripts/bench_decoder_step.py), persistent (default) vs MTTS_PDEC_LT=1 (per-step schedule above 128 characters)"
  for L in 120 128 160 200 256 304; do
    for lt in 3 1; do
      echo -n "L=$L MTTS_PDEC_LT=$lt  "; MTTS_PDEC_LT=$lt timeout 200 python scripts/bench_decoder_step.py --preset shared_training --batch 64 --chars $L --frames 300 2>&1 | tail -1
    done
  done
} > $O/long_inputs_ab.txt 2>&1
cat $O/long_inputs_ab.txt
timeout 1500 bash scripts/pmc_mfma.sh gpurun_out/r05b/pmc_mfma
