import sys, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from tests.helpers import build_hip_model, golden_names, hip_forward, load_golden
for name in golden_names('train'):
    fx = load_golden(name)
    model = build_hip_model(fx)
    post, pre, stop, align, spk, enc = hip_forward(fx, model)
    errs = {k: (v.cpu() - fx[k]).abs().max().item() for k, v in (('enc', enc), ('align', align), ('pre', pre), ('post', post), ('stop', stop)) if True for k2 in [k] for _ in [0] if (k in fx or True)} if False else None
    e = lambda a, b: (a.cpu() - b).abs().max().item()
    print(name, 'enc %.2e align %.2e pre %.2e post %.2e stop %.2e' % (e(enc, fx['encoder_output']), e(align, fx['alignment']), e(pre, fx['pre']), e(post, fx['post']), e(stop, fx['stop'])))
