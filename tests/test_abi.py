"""C-ABI checks that need no GPU: the library builds/loads, exports every declared symbol, and the ctypes
struct mirrors generated from include/mtts.h have the sizes the compiler gave them."""
import ctypes
import os

import pytest

from multilingual_text_to_speech_amd import _C


@pytest.fixture(scope='module')
def library():
    if not os.path.exists(_C.LIB_PATH):
        from multilingual_text_to_speech_amd.build import build
        build()
    return _C.lib()


def test_every_declared_symbol_is_exported(library):
    assert len(_C.FUNCTIONS) >= 15
    for name in _C.FUNCTIONS:
        assert hasattr(library, name), f'{name} declared in include/mtts.h but not exported'


def test_struct_mirrors_match_compiler_layout(library):
    order = ['GemmArgs', 'BnArgs', 'SkSeg', 'SkinnyArgs', 'AttnStepArgs', 'DecoderArgs', 'BiLstmArgs', 'AttnBwdArgs', 'DecoderGradArgs', 'BiLstmGradArgs', 'TacoLossArgs', 'AdamArgs', 'LstmPackArgs', 'LstmStepArgs', 'GenParamsArgs']
    for i, name in enumerate(order):
        assert library.mtts_sizeof_struct(i) == ctypes.sizeof(_C.STRUCTS[name]), name
    assert library.mtts_sizeof_struct(len(order)) == -1


def test_version_and_error_channel(library):
    assert library.mtts_version() >= 100
    assert isinstance(library.mtts_last_error(), bytes)


def test_no_cpu_fallback():
    """CPU tensors must be rejected loudly, not routed to some other implementation."""
    import torch
    from multilingual_text_to_speech_amd import kernels as K
    with pytest.raises(_C.MttsError):
        K.linear(torch.zeros(2, 4), torch.zeros(3, 4))
