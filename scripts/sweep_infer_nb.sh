for nb in 0 4 7; do
  if [ $nb = 0 ]; then unset MTTS_LS_NB; else export MTTS_LS_NB=$nb; fi
  timeout 300 python scripts/bench_inference.py --repeats 4 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('inference NB', '$nb', d['value'], 'frames/s', d['us_per_decoder_step'], 'us/step')"
done
