"""The attention+decoder forward step against the HBM roofline for any preset / batch / dtype - the function bench.py
puts on its line as `roofline` / `roofline_b240`:   python scripts/bench_decoder_step.py [--batch 240] [--preset generated_switching]
[--frames 300] [--dtype f32|bf16]"""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=240)
    ap.add_argument('--preset', default='generated_switching')
    ap.add_argument('--frames', type=int, default=300)
    ap.add_argument('--chars', type=int, default=120)
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16'])
    args = ap.parse_args()
    out = bench.secondary_step_roofline(args.preset, args.batch, args.chars, args.frames, torch.device('cuda', 0), args.dtype)
    print(json.dumps({k: out[k] for k in ('us_per_step', 'frac', 'achieved', 'bytes_per_step', 'dtype')} | {'batch': args.batch, 'preset': args.preset}))


if __name__ == '__main__':
    main()
