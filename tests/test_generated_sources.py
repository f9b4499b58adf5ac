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
