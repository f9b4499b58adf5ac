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


def test_buffer_size_queries_cover_every_caller_allocated_buffer(library):
    """mtts_decoder_buffer_elems / mtts_decoder_grad_buffer_elems / mtts_bilstm_buffer_elems: every pointer member of the argument
    blocks that the CALLER allocates (inputs, weights and gradient outputs aside) has a size query, the values follow the layouts in
    the header comments, unknown names return -1."""
    for f in ('mtts_decoder_buffer_elems', 'mtts_decoder_grad_buffer_elems', 'mtts_bilstm_buffer_elems'):
        getattr(library, f).restype = ctypes.c_long
    a = _C.DecoderArgs()
    a.B, a.L, a.T, a.M, a.P, a.H, a.A, a.Dm, a.ksz, a.C, a.n_prenet, a.kq, a.fast = 64, 120, 600, 80, 256, 1024, 128, 544, 31, 32, 2, 8, 1
    q = lambda f: library.mtts_decoder_buffer_elems(ctypes.byref(a), f.encode())
    assert q('h_att') == 601 * 64 * 1024 and q('ctx') == 601 * 64 * 544 and q('out') == 601 * 64 * 84 and q('align') == 600 * 64 * 120
    assert q('att_w2p') == 4 * 1024 * (544 + 1024) * 4 and q('qpart') == 64 * 64 * 128
    assert q('persist_ws') == library.mtts_decoder_persist_ws_bytes(64, 120, 1024, 544, 128) > 0
    assert q('no_such_field') == -1
    inputs = {'memory', 'lengths', 'frames_in', 'teacher', 'prenet_w', 'prenet_b', 'prenet_mask', 'att_hmask', 'att_cmask', 'gen_hmask', 'gen_cmask',
              'prenet_wp', 'prenet_act', 'persist_err'}
    weights = {n for n, _ in _C.DecoderArgs._fields_ if n.startswith(('att_w_', 'att_b_', 'gen_w_', 'gen_b_', 'w_', 'b_')) and not n.endswith(('_p', '_u', '2p'))}
    weights |= {'att_bias'}
    for name, ctype in _C.DecoderArgs._fields_:
        if ctype is ctypes.c_void_p and name not in inputs and name not in weights:
            assert q(name) > 0, name
    assert q('prenet_act') == 600 * 64 * 256 and q('prenet_wp0') == 256 * 80 and q('prenet_mask') == 600 * 64 * 256
    g = _C.DecoderGradArgs()
    g.ksb, g.ksb_ctx, g.nch = 4, 7, 4
    g.part_ring_slots = library.mtts_decoder_bwd_ring_slots()
    assert g.part_ring_slots >= 50
    qg = lambda f: library.mtts_decoder_grad_buffer_elems(ctypes.byref(a), ctypes.byref(g), f.encode())
    assert qg('part_att') == 7 * 64 * 544 + 4 * 64 * 1024 and qg('dU_slab') == 64 * 4 * 128 * 31 and qg('dpren') == 2 * 600 * 64 * 256
    for name, ctype in _C.DecoderGradArgs._fields_:
        if ctype is ctypes.c_void_p and not name.startswith('d_'):
            assert qg(name) > 0, name
    assert qg('prenet_w_T0') == 80 * 256 and qg('bogus') == -1
    b = _C.BiLstmArgs()
    b.B, b.L, b.Cin, b.H = 64, 120, 512, 256
    qb = lambda f: library.mtts_bilstm_buffer_elems(ctypes.byref(b), 4, f.encode())
    assert qb('h') == 121 * 64 * 256 and qb('part') == 2 * 4 * 64 * 256 and qb('y') == 64 * 120 * 512 and qb('zzz') == -1


def test_library_reports_no_debug_switches(library):
    assert library.mtts_build_flags() == 0
