"""Summarise the basic blocks of one kernel in a hipcc -S dump: python tools/isa_blocks.py file.s kernel_substring [--seq]"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
start = [i for i, l in enumerate(lines) if re.match(r'^_Z.*' + sys.argv[2] + '.*:', l)][0]
blocks, cur, name = [], [], 'entry'
for l in lines[start:]:
    if re.match(r'^\.LBB\d+_\d+:', l):
        blocks.append((name, cur)); name = l.split(':')[0]; cur = []
    else:
        cur.append(l)
    if l.strip().startswith('s_endpgm'):
        break
blocks.append((name, cur))
for name, b in blocks:
    ins = [x for x in b if x.strip() and not x.strip().startswith(';') and not x.strip().startswith('.')]
    nm = sum('v_mfma' in x for x in ins)
    sc = sum('scratch_' in x for x in ins)
    if nm or sc or len(ins) > 200:
        print(name, len(ins), 'mfma', nm, 'dsr', sum('ds_read' in x for x in ins), 'dsw', sum('ds_write' in x for x in ins), 'gl', sum('global_load' in x for x in ins),
              'bar', sum('s_barrier' in x for x in ins), 'scratch', sc, 'waits', sum('s_waitcnt' in x for x in ins))
    if nm >= 32 and '--seq' in sys.argv:
        seq = []
        for x in ins:
            t = x.strip().split()[0]
            if 's_waitcnt' in x: seq.append('W(' + x.strip().split(None, 1)[1] + ')')
            elif 'v_mfma' in x: seq.append('M')
            elif 'ds_read' in x: seq.append('r')
            elif 'ds_write' in x: seq.append('w')
            elif 'global_load' in x: seq.append('G')
            elif 'scratch_' in x: seq.append('S')
            elif 's_barrier' in x: seq.append('BAR')
            elif t.startswith('v_'): seq.append('v')
            elif t.startswith('s_'): seq.append('s')
            else: seq.append('?')
        print(' '.join(seq))
