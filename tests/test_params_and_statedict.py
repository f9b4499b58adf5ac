"""CPU checks of the drop-in boundary: config surface and checkpoint (state_dict) layout."""
import json
import os
import subprocess
import sys

import pytest
import torch

from multilingual_text_to_speech_amd.params import Params as hp, presets, reset_defaults

REF = '/root/reference'
needs_ref = pytest.mark.skipif(not os.path.isdir(REF), reason='reference tree not present on this machine')


def test_params_roundtrip_and_symbols(tmp_path):
    reset_defaults()
    assert hp.symbols_count() == 70 and hp.batch_size == 52 and hp.encoder_type == 'simple'
    presets.apply('generated_switching')
    assert hp.symbols_count() + 3 == 114 and hp.language_number == 5 and hp.generator_dim == 10
    p = tmp_path / 'x.json'
    hp.save(str(p))
    d = json.load(open(p))
    reset_defaults()
    hp.load(str(p))
    assert hp.encoder_type == 'generated' and hp.state_dict() == d
    presets.apply('shared_training')
    assert hp.symbols_count() + 3 == 147 and hp.language_embedding_dimension == 32


@needs_ref
def test_defaults_and_presets_equal_reference():
    code = ("import sys, json; sys.path.insert(0, %r); from params.params import Params as R; "
            "print(json.dumps(R.state_dict(), ensure_ascii=False))" % REF)
    ref = json.loads(subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, check=True).stdout)
    reset_defaults()
    assert hp.state_dict() == ref
    for name, overrides in presets.PRESETS.items():
        with open(os.path.join(REF, 'params', name + '.json'), encoding='utf-8') as f:
            assert json.load(f) == overrides, name


@needs_ref
@pytest.mark.parametrize('preset', ['shared_training', 'generated_switching', 'separate_training'])
def test_state_dict_layout_equals_reference(preset):
    """Same keys and shapes as the reference model => reference checkpoints load unchanged."""
    code = ("import sys, json, torch; sys.path.insert(0, %r); import utils; from params.params import Params as hp; "
            "from modules.tacotron2 import Tacotron; hp.load(%r); hp.language_number = len(hp.languages) if hp.multi_language else 0; "
            "hp.speaker_number = 91 if hp.multi_speaker else 0; m = Tacotron(); "
            "print(json.dumps({k: list(v.shape) for k, v in m.state_dict().items()}))"
            % (REF, os.path.join(REF, 'params', preset + '.json')))
    ref = json.loads(subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, check=True, cwd=REF).stdout)
    from multilingual_text_to_speech_amd.modules.tacotron2 import Tacotron
    presets.apply(preset, speaker_number=91 if presets.PRESETS[preset].get('multi_speaker') else 0)
    ours = {k: list(v.shape) for k, v in Tacotron().state_dict().items()}
    assert ours == ref


def test_golden_state_dict_loads_strictly():
    from tests.helpers import golden_names, load_golden
    from multilingual_text_to_speech_amd.modules.tacotron2 import Tacotron
    for name in golden_names():
        fx = load_golden(name)
        reset_defaults()
        hp.load_state_dict(fx['hp'])
        Tacotron().load_state_dict(fx['state_dict'], strict=True)


def test_install_aliases_exposes_reference_import_names():
    import multilingual_text_to_speech_amd as mtts
    saved = {k: sys.modules.get(k) for k in ('params', 'params.params', 'modules', 'modules.tacotron2', 'utils')}
    try:
        for k in saved:
            sys.modules.pop(k, None)
        mtts.install_aliases()
        from params.params import Params
        from modules.tacotron2 import Tacotron
        import utils
        assert Params is hp and hasattr(utils, 'build_model') and Tacotron.__name__ == 'Tacotron'
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def test_fused_adam_loads_the_reference_two_group_state_dict():
    """The reference builds its encoder_optimizer variant with the prenet / attention tensors listed twice in group 0
    (train.py:261-270); its saved optimizer state must load into our de-duplicated FusedAdam (train.make_optimizer)."""
    import warnings
    import torch
    from multilingual_text_to_speech_amd.optim import FusedAdam
    torch.manual_seed(0)
    shared = [torch.nn.Parameter(torch.randn(3, 2)), torch.nn.Parameter(torch.randn(4))]
    other = [torch.nn.Parameter(torch.randn(5)), torch.nn.Parameter(torch.randn(2, 2))]
    enc = [torch.nn.Parameter(torch.randn(6))]
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        ref = torch.optim.Adam([{'params': shared + other + shared}, {'params': enc, 'lr': 3e-4}], lr=1e-3, weight_decay=1e-6)
    for p in shared + other + enc:
        p.grad = torch.randn_like(p)
    ref.step()
    sd = ref.state_dict()
    assert len(sd['param_groups'][0]['params']) == 6          # duplicated indices as the reference saves them
    mine = FusedAdam([{'params': shared + other}, {'params': enc, 'lr': 3e-4}], lr=1e-3, weight_decay=1e-6)
    mine.load_state_dict(sd)
    assert [len(g['params']) for g in mine.param_groups] == [4, 1]
    assert mine.param_groups[1]['lr'] == 3e-4
    for p in shared + other + enc:
        assert torch.equal(mine.state[p]['exp_avg'], ref.state[p]['exp_avg'])
        assert float(mine.state[p]['step']) == float(ref.state[p]['step'])      # (torch steps a twice-listed tensor twice)


def test_models_outside_the_persistent_kernels_geometry_say_so_once():
    """decoder_dimension 1024 / attention_dimension 128 / memory width % 32 are what the persistent decoder kernels are laid out for
    (csrc/persist.hip); other models run the per-step schedule at ~2.5x the decoder time - with ONE warning per shape, not silently."""
    import warnings
    from multilingual_text_to_speech_amd.modules import tacotron2
    from multilingual_text_to_speech_amd.params import presets
    tacotron2._WARNED_SHAPES.clear()
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter('always')
        presets.apply('shared_training')
        tacotron2.Tacotron()
        assert not [c for c in caught if 'persistent decoder kernels' in str(c.message)]
        presets.apply('shared_switching', speaker_number=7)           # memory width 256 + 4 + 32 = 292
        tacotron2.Tacotron()
        tacotron2.Tacotron()
        msgs = [str(c.message) for c in caught if 'persistent decoder kernels' in str(c.message)]
    assert len(msgs) == 1 and '292' in msgs[0]


def test_settle_host_heap_freezes_the_long_lived_objects(monkeypatch):
    """utils.settle_host_heap: one full collection, survivors into the collector's permanent generation (the first generation-2 pass of
    a training run otherwise costs one whole train step of GPU idle time: profiles/r06_host_gc_outlier.txt); MTTS_HOST_GC_FREEZE=0 = off;
    bench.py calls it between the warm-up and the timed steps, train.py two steps into a run / an epoch."""
    import gc
    from multilingual_text_to_speech_amd.utils import settle_host_heap
    gc.unfreeze()
    monkeypatch.setenv('MTTS_HOST_GC_FREEZE', '0')
    assert settle_host_heap() == 0 and gc.get_freeze_count() == 0
    monkeypatch.setenv('MTTS_HOST_GC_FREEZE', '1')
    n = settle_host_heap()
    assert n > 1000 and gc.get_freeze_count() == n and gc.isenabled()
    gc.unfreeze()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, 'bench.py')).read()
    assert src.index('settle_host_heap()') < src.index('t0 = time.perf_counter()\n    sync_each')        # before the timed region
    assert 'settle_host_heap()' in open(os.path.join(root, 'train.py')).read()
