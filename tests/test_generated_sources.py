"""Host-side build artefacts that no GPU is needed to check."""


def test_committed_gemm_stream_is_what_the_generator_emits():
    """csrc/gemm_pipe_body.inc (the hand-placed instruction stream of gemm_pipe_kernel) is generated: the committed file must equal the
    generator's output for its default settings, and hold exactly one MFMA per group and one block barrier."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'scripts', 'gen_gemm_pipe.py')], capture_output=True, text=True, check=True).stdout
    committed = open(os.path.join(root, 'multilingual_text_to_speech_amd', 'csrc', 'gemm_pipe_body.inc')).read()
    assert out == committed
    groups = [l for l in committed.splitlines() if l.startswith('PP_MFMA')]
    assert len(groups) == 48 and all(l.count('PP_MFMA(') == 1 and l.endswith('PP_SB') for l in groups)
    assert committed.count('PP_BARRIER') == 1
    # every fragment of both k-halves is read exactly once per block, every split stage of both operands appears once
    assert committed.count('PP_RDA(') == 12 and committed.count('PP_RDB(') == 12
    for o in 'AB':
        for n in range(8):
            for stage in ('S1A', 'S1B', 'S2A', 'S2B'):
                assert committed.count(f'PP_{stage}({o}, {n})') == 1


def test_committed_presplit_gemm_streams_are_what_the_generator_emits():
    """csrc/gemm_planes_body.inc / gemm_planes_body_bf16.inc (K step of gemm_planes_kernel, fp32 mode: 48 MFMAs of the six terms; bf16
    mode: 24 MFMAs, three K blocks per record) equal scripts/gen_gemm_planes.py's output; every step stores and requests the six
    quanta of both operands once, reads every fragment of both k-halves once and holds one block barrier."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for flag, name, n_mfma in ((None, 'gemm_planes_body.inc', 48), ('--bf16', 'gemm_planes_body_bf16.inc', 24)):
        cmd = [sys.executable, os.path.join(root, 'scripts', 'gen_gemm_planes.py')] + ([flag] if flag else [])
        out = subprocess.run(cmd, capture_output=True, text=True, check=True).stdout
        committed = open(os.path.join(root, 'multilingual_text_to_speech_amd', 'csrc', name)).read()
        assert out == committed, name
        groups = [l for l in committed.splitlines() if l.startswith('PL_MFMA')]
        assert len(groups) == n_mfma and all(l.count('PL_MFMA(') == 1 and l.endswith('PL_SB') for l in groups)
        assert committed.count('PL_BARRIER') == 1
        assert committed.count('PL_RDA(') == 12 and committed.count('PL_RDB(') == 12
        for o in 'AB':
            assert committed.count(f'PL_NEXT({o})') == 1
            for j in range(6):
                assert committed.count(f'PL_ST({o}, {j})') == 1 and committed.count(f'PL_LD({o}, {j})') == 1


def test_product_kernel_sources_carry_no_harness_instrumentation():
    """The stage clocks and stream knock-outs of the micro-benchmark harnesses live in scripts/mb/instrumentation/*.patch
    (scripts/mb/instrument.py), not behind #ifdef in the product's kernel sources - and the patches still apply to them."""
    import os, re, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, 'multilingual_text_to_speech_amd', 'csrc')
    pat = re.compile(r'LF_NO_|PS_PROF|PN_PROF|ATB_PROF|_STAMP\(|KO_LD|KO_MFMA')
    for f in sorted(os.listdir(csrc)):
        if f.endswith(('.hip', '.h', '.inc', '.cpp')):
            hits = [l for l in open(os.path.join(csrc, f)).read().splitlines() if pat.search(l)]
            assert not hits, (f, hits[:3])
    r = subprocess.run([sys.executable, os.path.join(root, 'scripts', 'mb', 'instrument.py'), 'apply'], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_gemm_tile_order_is_a_bijection_for_every_grid():
    """gemm_tile_block (csrc/gemm.hip): linear tile index -> (tile row, tile column) in column strips of four tiles.  The device function
    is restated here line by line; for every grid shape up to 40 x 40 each tile must be produced exactly once, and 32 consecutive
    indices of a full strip must cover 8 x 4 tiles (what an XCD's workgroups share in L2)."""
    import os, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, 'multilingual_text_to_speech_amd', 'csrc', 'gemm.hip')).read()
    body = src[src.index('void gemm_tile_block('):]
    body = body[:body.index('\n}\n')]
    # the restatement below mirrors these statements; if the device code changes, change both
    for stmt in ('constexpr int W = 4;', 'const int nfull = ntx / W, per_strip = W * nty;', 'int strip = id / per_strip, rem = id - strip * per_strip, w = W;',
                 'if (strip >= nfull) { strip = nfull; rem = id - nfull * per_strip; w = ntx - W * nfull; }', 'tile_m = rem / w;',
                 'tile_n = strip * W + (rem - tile_m * w);'):
        assert stmt in body, stmt

    def tile(id_, ntx, nty, W=4):
        nfull, per_strip = ntx // W, W * nty
        strip, rem, w = id_ // per_strip, id_ % per_strip, W
        if strip >= nfull:
            strip, rem, w = nfull, id_ - nfull * per_strip, ntx - W * nfull
        tm = rem // w
        return tm, strip * W + (rem - tm * w)

    for ntx in range(1, 41):
        for nty in range(1, 41):
            seen = {tile(i, ntx, nty) for i in range(ntx * nty)}
            assert len(seen) == ntx * nty and all(0 <= m < nty and 0 <= n < ntx for m, n in seen), (ntx, nty)
    block = {tile(i, 32, 300) for i in range(64, 96)}
    assert len({m for m, _ in block}) == 8 and len({n for _, n in block}) == 4
