"""Instruction mix of the largest MFMA-carrying loop of every kernel in a gfx950 assembly file (hipcc -S --cuda-device-only ...).
Looks for what the source does not show: accumulator copies around predicated MFMAs (v_mov / v_accvgpr), integer-division sequences,
IEEE division / denormal scaling around transcendentals.   usage: isa_loop_mix.py file.s [name substring]"""
import collections
import re
import subprocess
import sys

s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
kern = re.findall(r'^(_Z\w+):[^\n]*\n(.*?)\n\s*s_endpgm', s, re.S | re.M)


def demangle(n):
    try:
        return subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt', n], capture_output=True, text=True).stdout.strip()
    except Exception:
        return n


for name, body in kern:
    dn = demangle(name)
    if pat not in dn:
        continue
    body = body.split('\n')
    labels = {}
    for i, l in enumerate(body):
        mm = re.match(r'^(\.LBB\d+_\d+):', l)
        if mm:
            labels[mm.group(1)] = i
    best = None
    for i, l in enumerate(body):
        mm = re.search(r's_cbranch_\w+ (\.LBB\d+_\d+)', l) or re.search(r's_branch (\.LBB\d+_\d+)', l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            seg = body[labels[mm.group(1)]:i + 1]
            nm = sum('v_mfma' in x for x in seg)
            if nm and (best is None or len(seg) > len(best)):
                best = seg
    if not best:
        continue
    c = collections.Counter(l.split()[0] for l in best if l.strip() and not l.strip().startswith(('.', ';')) and not l.strip().endswith(':'))
    mf = sum(n for o, n in c.items() if 'mfma' in o)
    valu = sum(n for o, n in c.items() if o.startswith('v_') and 'mfma' not in o)
    mov = sum(n for o, n in c.items() if o.startswith(('v_mov', 'v_accvgpr', 'v_pk_mov')))
    div = sum(n for o, n in c.items() if o.startswith(('v_div', 'v_rcp_iflag', 'v_mul_hi_u32', 'v_ldexp', 'v_cmp_class')))
    salu = sum(n for o, n in c.items() if o.startswith('s_'))
    lds = sum(n for o, n in c.items() if o.startswith('ds_'))
    print(f"{dn.split('(')[0][:66]:66s} loop {len(best):5d}  mfma {mf:4d} valu {valu:5d} (mov {mov:4d}, div/scale {div:3d}) salu {salu:4d} lds {lds:4d}")
