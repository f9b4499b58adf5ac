"""MI355X-native multilingual Tacotron-2 text->mel engine.

    import multilingual_text_to_speech_amd as mtts
    mtts.install_aliases()            # optional: `from modules.tacotron2 import Tacotron`, `from params.params import Params`
"""
import os
import sys

# The HIP runtime reads this flag when it is loaded (with torch): it only takes effect if the package is imported before torch.
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')

__version__ = "0.1.0"


def install_aliases():
    """Expose the package's `params`, `modules`, `utils` under the reference's top-level import names."""
    from . import params, modules, utils
    from .params import params as _pp
    from .modules import tacotron2, encoder, attention, layers, generated, classifier
    sys.modules.setdefault('params', params)
    sys.modules.setdefault('params.params', _pp)
    sys.modules.setdefault('modules', modules)
    for m in (tacotron2, encoder, attention, layers, generated, classifier):
        sys.modules.setdefault('modules.' + m.__name__.rsplit('.', 1)[1], m)
    sys.modules.setdefault('utils', utils)
