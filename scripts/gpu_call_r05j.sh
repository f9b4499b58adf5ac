# NOTE: the A/B switch MTTS_ATTN_BIG_NCH measured by the second half of this call was removed from the product afterwards (result: profiles/r05_attn_big_nch.txt)
# round 5, call J: A/Bs - long inputs old vs new schedule (forward + train step), large-batch attention with two workgroups per sample
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r05j; mkdir -p $O
{
echo "# scripts/ab_long_inputs.py: shared_training, batch 64, L = 200 (ragged U[100,200]), T = 300 / 600: decoder forward us per step, whole train step"
for T in 300 600; do
echo -n "T=$T round-5 schedule (persistent LT=2, attention backward nch = L/32): "; timeout 300 python scripts/ab_long_inputs.py $T 2>/dev/null | tail -1
echo -n "T=$T round-4 schedule (MTTS_PDEC_LT=1 MTTS_NCH_BWD=4):                  "; MTTS_PDEC_LT=1 MTTS_NCH_BWD=4 timeout 300 python scripts/ab_long_inputs.py $T 2>/dev/null | tail -1
echo -n "T=$T persistent forward, old backward (MTTS_NCH_BWD=4):                "; MTTS_NCH_BWD=4 timeout 300 python scripts/ab_long_inputs.py $T 2>/dev/null | tail -1
done
} > $O/long_inputs_train_ab.txt 2>&1
cat $O/long_inputs_train_ab.txt
{
for dt in f32 bf16; do for n in 1 2; do echo -n "b240 $dt MTTS_ATTN_BIG_NCH=$n: "; MTTS_ATTN_BIG_NCH=$n timeout 200 python scripts/bench_decoder_step.py --batch 240 --dtype $dt 2>/dev/null | tail -1; done; done
} > $O/attn_big_nch.txt 2>&1
cat $O/attn_big_nch.txt
