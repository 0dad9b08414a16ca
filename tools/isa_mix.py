"""Instruction mix per kernel of an AMDGPU assembly file (hipcc -S --cuda-device-only): counts by class, and the
same restricted to the hottest loop (largest backward-branch span)."""
import collections, re, sys

def classify(i):
    if i.startswith('v_fmac_f32_dpp'): return 'fmac_dpp'
    if i.startswith('v_mfma'): return 'mfma'
    if i.startswith('v_accvgpr'): return 'accvgpr'
    if i.startswith('v_') and ('_dpp' in i): return 'other_dpp'
    if i.startswith('v_exp') or i.startswith('v_rcp') or i.startswith('v_rsq') or i.startswith('v_sqrt'): return 'trans'
    if i.startswith('v_'): return 'valu'
    if i.startswith('s_waitcnt'): return 'waitcnt'
    if i.startswith('s_nop'): return 'nop'
    if i.startswith('s_'): return 'salu'
    if i.startswith('ds_'): return 'ds'
    if i.startswith('scratch_'): return 'scratch'
    if i.startswith('global_') or i.startswith('buffer_') or i.startswith('flat_'): return 'vmem'
    return 'other'

lines = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2] if len(sys.argv) > 2 else ''
name = None
body = []
def report(name, body):
    c = collections.Counter()
    for l in body:
        t = l.strip()
        if not t or t.startswith('.') or t.startswith(';') or t.endswith(':'): continue
        # an asm block line may hold several instructions
        c[classify(t.split()[0])] += 1
    print(name[:110], sum(c.values()), dict(sorted(c.items())))
for l in lines:
    m = re.match(r'(_Z\S+):', l)
    if m:
        if name and pat in name: report(name, body)
        name, body = m.group(1), []
    elif l.startswith('\t.end_amdhsa_kernel') or l.startswith('.Lfunc_end'):
        if name and pat in name: report(name, body)
        name, body = None, []
    elif name is not None:
        body.append(l)
