"""Static instruction-class table of a kernel from the compiler's ISA (tools/kinfo.sh leaves it in /tmp/dis):
whole kernel, and every straight-line stretch of at least MIN instructions (the loop bodies: the stream waves'
two pieces of a unit, the service waves' tile recurrence ...).  usage: python tools/isa_classes.py <kernel regex> [MIN]"""
import collections, re, sys
txt = open('/tmp/dis/hrfd_lib-hip-amdgcn-amd-amdhsa-gfx950.s').read()
pat = re.compile(sys.argv[1]); MIN = int(sys.argv[2]) if len(sys.argv) > 2 else 150


def cls(m, ops):
    if m.startswith(('buffer_', 'global_', 'flat_', 'scratch_')): return 'VMEM ' + ('load' if 'load' in m else 'store' if 'store' in m else 'atomic')
    if m.startswith('ds_'): return 'LDS'
    if m.startswith('s_waitcnt'): return 's_waitcnt'
    if m.startswith('s_nop'): return 's_nop'
    if m.startswith(('s_cbranch', 's_branch')): return 'branch'
    if m.startswith('s_'): return 'SALU/SMEM'
    if m.startswith('v_'):
        if 'dpp' in ops or m.endswith('_dpp'): return 'VALU DPP'
        if 'sdwa' in ops or m.endswith('_sdwa'): return 'VALU SDWA'
        if m.startswith('v_pk_'): return 'VALU packed (VOP3P)'
        if m.startswith(('v_perm', 'v_lerp', 'v_dot2', 'v_alignb', 'v_bfe', 'v_bfi', 'v_add3', 'v_lshl_add', 'v_and_or', 'v_or3', 'v_med3', 'v_max3', 'v_min3',
                         'v_mad_', 'v_fma_f32', 'v_bitop3', 'v_lshl_or', 'v_xad', 'v_add_lshl', 'v_cndmask_b32_e64', 'v_readlane', 'v_writelane', 'v_mbcnt', 'v_cvt_pk')) or m.endswith('_e64'):
            return 'VALU 64-bit encoding (VOP3)'
        if m.startswith('v_cmp'): return 'VALU compare'
        if m.startswith(('v_rcp', 'v_rsq', 'v_sqrt', 'v_exp', 'v_log', 'v_sin', 'v_cos')): return 'VALU transcendental'
        return 'VALU 32-bit encoding (VOP1/VOP2)'
    return 'other'


for name in re.findall(r'^(_Z\w+):', txt, re.M):
    if not pat.search(name):
        continue
    a = txt.index(name + ':'); b = txt.index('.end_amdhsa_kernel', a) if '.end_amdhsa_kernel' in txt[a:] else len(txt)
    body = txt[a:txt.index('s_endpgm', a) if 's_endpgm' in txt[a:b] else b]
    body = txt[a:b]
    lines = [l.strip() for l in body.split('\n')]
    ins = []
    for l in lines:
        if not l or l.startswith((';', '.', '_Z')) or l.endswith(':'):
            ins.append(('LABEL', l)) if l.endswith(':') else None
            continue
        parts = l.split(None, 1)
        ins.append((parts[0], parts[1] if len(parts) > 1 else ''))
    def table(seq, title):
        c = collections.Counter(cls(m, o) for m, o in seq if m != 'LABEL')
        n = sum(c.values())
        print(f"{title}: {n} instructions")
        for k, v in sorted(c.items(), key=lambda kv: -kv[1]):
            print(f"    {k:34s} {v:6d}  {100.0 * v / n:5.1f} %")
    print("==", name)
    table(ins, "whole kernel (static)")
    run, start = [], 0
    for i, (m, o) in enumerate(ins + [('LABEL', '')]):
        if m == 'LABEL' or m.startswith(('s_cbranch', 's_branch', 's_endpgm')):
            if len(run) >= MIN:
                head = next((l for l in run[:1]), None)
                marks = [o for mm, o in run if mm.startswith('buffer_load')]
                lds = sum(1 for mm, o in run if mm.startswith('ds_'))
                table(run, f"straight-line stretch #{start} ({len(marks)} buffer loads, {lds} LDS operations)")
            run, start = [], i + 1
        else:
            run.append((m, o))
