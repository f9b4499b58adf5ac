# attention kernel: workgroups per sample (MTTS_ATTN_NCH) vs decoder forward step time at several batch sizes
for cfg in "shared_training 64" "generated_switching 240" "generated_switching 120"; do
  set -- $cfg
  for nch in 2 3 4; do
    MTTS_ATTN_NCH=$nch python scripts/bench_decoder_step.py --preset $1 --batch $2 --frames 200 2>/dev/null | tail -1
  done
done
