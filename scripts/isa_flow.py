"""Developer aid: compressed instruction flow (waits, MFMA runs, memory ops, branches) of one kernel
in a hipcc -S dump.  usage: isa_flow.py file.s <mangled-name-prefix>"""
import sys
lines = open(sys.argv[1]).read().splitlines()
pre = sys.argv[2]
st = next(i for i, l in enumerate(lines) if l.startswith(pre) and ':' in l)
en = next(i for i in range(st, len(lines)) if '.end_amdhsa_kernel' in lines[i])
prev, cnt, out = None, 0, []
for l in lines[st:en]:
    t = l.strip()
    if not l.startswith('\t'):
        if t.startswith('.LBB'):
            c = '\n' + t.split(':')[0] + ':'
        else:
            continue
    elif t.startswith(('.', ';')):
        continue
    else:
        op = t.split()[0]
        if op == 's_waitcnt' or op.startswith(('s_cbranch', 's_branch')):
            c = t.split(';')[0].strip()
        elif op.startswith(('v_mfma', 'global_', 'v_mov_b64', 'ds_', 's_atomic', 's_barrier', 'scratch', 's_sleep')):
            c = op
        else:
            c = '.'
    if c == prev:
        cnt += 1
    else:
        if prev:
            out.append(prev + (' x%d' % cnt if cnt > 1 else ''))
        prev, cnt = c, 1
out.append(prev)
print(' | '.join(out))
