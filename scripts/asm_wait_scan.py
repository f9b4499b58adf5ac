"""Static scan of the gfx950 assembly of the library's kernels for SERIALISED memory round trips.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -x hip --cuda-device-only -S -o x.s csrc/x.hip
    python scripts/asm_wait_scan.py x.s [kernel-name-substring]

A latency-bound step kernel should issue its global loads in a burst and wait once.  What this flags per kernel: every
`s_waitcnt vmcnt(N)` that leaves at most N loads in flight while fewer than 3 vector-memory loads were issued since the previous
such wait ("short round trip") - the signature of `cond ? *p : 0` loads and of loads under per-load branches, where the compiler
waits for each value right behind its load (round 4: skinny row-major operands, prenet2's keep flags).  Straight-line count only:
waits inside a loop body are listed once, so read the numbers as a map of where to look, not as a cycle model.
"""
import re
import sys


def kernels(path):
    name, body = None, []
    for line in open(path):
        m = re.match(r'^(_Z\w+|\w+_kernel\w*):\s', line)
        if m and not line.startswith('.'):
            if name:
                yield name, body
            name, body = m.group(1), []
        elif name is not None:
            body.append(line)
            if 's_endpgm' in line:
                yield name, body
                name, body = None, []


def scan(body):
    loads_since, n_loads, short, waits0 = 0, 0, 0, 0
    seq = []
    for line in body:
        t = line.split()
        if not t:
            continue
        op = t[0]
        if op.startswith(('global_load', 'flat_load', 'buffer_load', 'scratch_load')):
            loads_since += 1
            n_loads += 1
        elif op == 's_waitcnt' and 'vmcnt' in line:
            m = re.search(r'vmcnt\((\d+)\)', line)
            n = int(m.group(1))
            if n == 0:
                waits0 += 1
            if n <= 1 and 0 < loads_since < 3:
                short += 1
            seq.append((loads_since, n))
            if n <= 1:
                loads_since = 0
    return n_loads, waits0, short, seq


def stats(body):
    """Instruction statistics of the straight-line listing: total, vector ALU, scalar ALU, MFMA, LDS, quarter-rate integer multiplies /
    64-bit multiply-adds, IEEE divisions, SGPR-spill lane moves."""
    ops = [l.split()[0] for l in body if l.startswith('\t') and l.split() and not l.split()[0].startswith(('.', ';'))]
    return dict(instr=len(ops), valu=sum(o.startswith('v_') for o in ops), salu=sum(o.startswith('s_') for o in ops),
                mfma=sum('mfma' in o for o in ops), lds=sum(o.startswith('ds_') for o in ops),
                mul=sum(o.startswith(('v_mul_lo', 'v_mul_hi', 'v_mad_u64', 'v_mad_i64')) for o in ops), div=sum(o == 'v_div_scale_f32' for o in ops),
                lane=sum(o in ('v_readlane_b32', 'v_writelane_b32') for o in ops))


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == '--table':
    # python scripts/asm_wait_scan.py --table a.s b.s ...: one line per kernel with instruction statistics + the wait summary
    print(f'{"kernel":64s} {"instr":>6s} {"valu":>5s} {"salu":>5s} {"mfma":>5s} {"lds":>4s} {"mul64":>5s} {"div":>4s} {"lane":>5s} | {"loads":>5s} {"wait0":>5s} {"short":>5s} {"burst":>5s}')
    for path in sys.argv[2:]:
        for name, body in kernels(path):
            if not any(l.split()[:1] == ['s_endpgm'] for l in body):
                continue
            st = stats(body)
            n_loads, waits0, short, seq = scan(body)
            print(f'{name[:64]:64s} {st["instr"]:6d} {st["valu"]:5d} {st["salu"]:5d} {st["mfma"]:5d} {st["lds"]:4d} {st["mul"]:5d} {st["div"]:4d} {st["lane"]:5d} | '
                  f'{n_loads:5d} {waits0:5d} {short:5d} {seq[0][0] if seq else 0:5d}')
    sys.exit(0)

if __name__ == '__main__':
    path = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else ''
    print(f'{"kernel":72s} {"loads":>6s} {"vmcnt(0)":>9s} {"short round trips":>18s}')
    for name, body in kernels(path):
        if pat and pat not in name:
            continue
        n_loads, waits0, short, seq = scan(body)
        print(f'{name[:72]:72s} {n_loads:6d} {waits0:9d} {short:18d}')
        if pat:
            print('   (loads issued since the last full wait, vmcnt) :', ' '.join(f'{a}/{b}' for a, b in seq))
