"""Shared test helpers: fixture loading and oracle replay."""
import glob
import os

import torch

from oracle import tacotron_oracle as O

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


def golden_names(kind=None):
    names = sorted(os.path.basename(p)[:-3] for p in glob.glob(os.path.join(GOLDEN, '*.pt')))
    if kind == 'train':
        return [n for n in names if not n.endswith('_infer')]
    if kind == 'infer':
        return [n for n in names if n.endswith('_infer')]
    return names


def load_golden(name):
    return torch.load(os.path.join(GOLDEN, name + '.pt'), weights_only=False)


def cfg_of(fx):
    """Oracle cfg dict = reference defaults + the fixture's overrides."""
    from multilingual_text_to_speech_amd.params import Params, reset_defaults
    reset_defaults()
    Params.load_state_dict(fx['hp'])
    cfg = O.cfg_from_params(Params)
    reset_defaults()
    return cfg


def oracle_replay(fx, with_grads=False):
    cfg = cfg_of(fx)
    sd = {k: v.clone() for k, v in fx['state_dict'].items()}
    if with_grads:
        for k, v in sd.items():
            if v.is_floating_point() and not k.endswith(('running_mean', 'running_var')):
                v.requires_grad_(True)
    out = O.tacotron_forward(sd, cfg, fx['text'], fx['text_length'], fx['target'], fx['target_length'],
                             fx['speakers'], fx['languages'], fx['teacher'], fx['masks'], fx['train'])
    loss, parts = O.tacotron_loss(cfg, out, fx['text_length'], fx['target_length'], fx['target'],
                                  fx['stop_target'], fx['speakers'], fx['guided_g'])
    grads = None
    if with_grads:
        loss.backward()
        grads = {k: v.grad for k, v in sd.items() if v.requires_grad and v.grad is not None}
    return out, loss, parts, grads
