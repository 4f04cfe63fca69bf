"""Rough VGPR liveness over the hottest loop of one kernel in a hipcc -S dump (inner branches treated as fall-through):
python tools/isa_liveness.py file.s kernel_substring loop_label  -> pressure profile and the long-lived registers."""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
start = [i for i, l in enumerate(lines) if re.match(r'^_Z.*' + sys.argv[2] + '.*:', l)][0]
lab = sys.argv[3]
i0 = next(i for i in range(start, len(lines)) if lines[i].startswith(lab + ':'))
i1 = max(i for i in range(i0, len(lines)) if re.search(r's_cbranch\w+\s+' + re.escape(lab) + r'\b', lines[i]) or re.search(r's_branch\s+' + re.escape(lab) + r'\b', lines[i]))
body = [l.split(';')[0].strip() for l in lines[i0 + 1:i1 + 1]]
body = [l for l in body if l and not l.startswith('.') and not l.endswith(':')]
def regs(tok):
    out = []
    for m in re.finditer(r'\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b', tok):
        if m.group(1): out += [m.group(1) + str(k) for k in range(int(m.group(2)), int(m.group(3)) + 1)]
        else: out.append(m.group(4) + m.group(5))
    return out
NODEF = ('buffer_store', 'global_store', 'ds_write', 'scratch_store', 's_', 'v_cmp', 'v_cmpx', 'ds_bpermute_x')
ins = []
for l in body:
    op, _, rest = l.partition(' ')
    toks = [t.strip() for t in rest.split(',')]
    if op.startswith(NODEF) and not op.startswith('v_cmp') or op.startswith('s_'):
        d, u = [], sum((regs(t) for t in toks), [])
    elif op.startswith('v_cmp'):
        d, u = [], sum((regs(t) for t in toks), [])
    else:
        d = regs(toks[0]) if toks else []
        u = sum((regs(t) for t in toks[1:]), [])
        if op.startswith(('v_fmac', 'v_mac', 'v_pk_fmac')) or 'dpp' in l or 'v_mfma' in op and toks[-1].startswith(('a[', 'v[')): u += [] if 'v_mfma' in op else d
    ins.append((l, set(d), set(u)))
live = set()
for _ in range(3):
    prof = []
    for l, d, u in reversed(ins):
        live = (live - d) | u
        prof.append(len(live))
prof.reverse()
print(f'{len(ins)} instructions, live at loop top {len(live)}, max {max(prof)} at #{prof.index(max(prof))}: {ins[prof.index(max(prof))][0]}')
step = max(1, len(ins) // 40)
for k in range(0, len(ins), step): print(f'{k:5d} {prof[k]:4d}  {ins[k][0][:70]}')
