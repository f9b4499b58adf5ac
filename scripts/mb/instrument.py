"""Timing instrumentation of the micro-benchmark harnesses lives HERE, not in the product sources.

The stage clocks (PS_PROF / PN_PROF / ATB_PROF / PB_PROF: clock stamps between the stages of a kernel; PB_PROF = the persistent decoder
backward of round 6: build the library with MTTS_EXTRA_FLAGS=-DPB_PROF from the instrumented copy and run with MTTS_PBWD=1 MTTS_PBWD_CLOCK=1) and the stream knock-outs (LF_NO_X /
_W / _MFMA / _EPI / _STAGE: what a fused LSTM step costs without one of its streams) used to sit behind #ifdef in csrc/persist.hip,
csrc/lstm_step.hip and csrc/attention_bwd_body.h.  They are now patches under scripts/mb/instrumentation/:

    python scripts/mb/instrument.py apply     # csrc/<file> + instrumentation/<file>.patch -> scripts/mb/gen/<file> (what the harnesses include)
    python scripts/mb/instrument.py strip     # (maintenance) instrumented sources in csrc/ -> stripped sources + regenerated patches
    python scripts/mb/instrument.py repatch   # (maintenance) csrc/<file> edited by hand: regenerate the patch against gen/<file>.instrumented

`apply` fails loudly when a patch no longer fits the product source."""
import os, re, subprocess, sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, '..', '..', 'multilingual_text_to_speech_amd', 'csrc')
FILES = ['persist.hip', 'lstm_step.hip', 'attention_bwd_body.h', 'pbwd.hip']
COPIES = ['attention_bwd.hip']      # sources that only INCLUDE an instrumented file: copied next to it so that the include resolves to the copy
SYMS = ('LF_NO_X', 'LF_NO_W', 'LF_NO_MFMA', 'LF_NO_EPI', 'LF_NO_STAGE', 'PS_PROF', 'PN_PROF', 'ATB_PROF', 'PB_PROF')
STAMP = re.compile(r'^\s*(PD_STAMP|PN_STAMP|ATB_STAMP|PS_STAMP|PB_STAMP)\(.*\);\s*(//.*)?$')
STAMP_DEF = re.compile(r'^\s*#\s*(define|undef)\s+(PD_STAMP|PN_STAMP|ATB_STAMP|PS_STAMP|PB_STAMP)\b')


def strip_text(text):
    out, stack = [], []           # stack entries: None (foreign conditional) or [keep_now]
    for line in text.split('\n'):
        s = line.strip()
        m = re.match(r'#\s*(ifdef|ifndef|if)\b(.*)', s)
        if m:
            kind, rest = m.group(1), m.group(2).split('//')[0].strip()
            names = set(re.findall(r'[A-Za-z_][A-Za-z0-9_]*', rest)) - {'defined'}
            if names and names <= set(SYMS):
                keep = kind == 'ifndef'                    # every harness symbol is undefined in the product
                stack.append([keep])
                continue
            stack.append(None)
        elif re.match(r'#\s*else\b', s) and stack and stack[-1] is not None:
            stack[-1][0] = not stack[-1][0]
            continue
        elif re.match(r'#\s*endif\b', s) and stack:
            top = stack.pop()
            if top is not None:
                continue
        if all(e is None or e[0] for e in stack):
            if STAMP.match(line) or STAMP_DEF.match(line):
                continue
            out.append(line)
    return '\n'.join(out)


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else 'apply'
    gen = os.path.join(HERE, 'gen')
    os.makedirs(gen, exist_ok=True)
    for f in FILES:
        src = os.path.join(CSRC, f)
        patch = os.path.join(HERE, 'instrumentation', f + '.patch')
        if mode == 'repatch':      # (maintenance) product source edited by hand: csrc/<file> vs gen/<file>.instrumented -> patch
            r = subprocess.run(['diff', '-u', '--label', f, '--label', f, src, os.path.join(gen, f + '.instrumented')], capture_output=True, text=True)
            open(patch, 'w').write(r.stdout)
            print(f'{f}: patch of {len(r.stdout.splitlines())} lines')
            continue
        if mode == 'strip':
            text = open(src).read()
            inst = os.path.join(gen, f + '.instrumented')
            open(inst, 'w').write(text)
            open(src, 'w').write(strip_text(text))
            r = subprocess.run(['diff', '-u', '--label', f, '--label', f, src, inst], capture_output=True, text=True)
            open(patch, 'w').write(r.stdout)
            print(f'{f}: stripped, patch of {len(r.stdout.splitlines())} lines')
        else:
            dst = os.path.join(gen, f)
            open(dst, 'w').write(open(src).read())
            r = subprocess.run(['patch', '--no-backup-if-mismatch', '-s', dst, patch], capture_output=True, text=True)
            if r.returncode != 0:
                raise SystemExit(f'{f}: the instrumentation patch no longer applies:\n{r.stdout}{r.stderr}')
            print(f'{f}: instrumented copy -> {os.path.relpath(dst)}')


def copies():
    for f in COPIES:
        open(os.path.join(HERE, 'gen', f), 'w').write(open(os.path.join(CSRC, f)).read())


if __name__ == '__main__':
    main()
    if (sys.argv[1] if len(sys.argv) > 1 else 'apply') == 'apply':
        copies()
