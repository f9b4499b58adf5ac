for nb in 4 7 10; do for dt in f32 bf16; do
  MTTS_LS_NB=$nb python scripts/bench_decoder_step.py --preset generated_switching --batch 240 --frames 300 --dtype $dt 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('NB', $nb, d['dtype'], 'us/step', d['us_per_step'], 'frac', d['frac'])"
done; done
