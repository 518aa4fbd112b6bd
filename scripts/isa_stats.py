"""Developer aid: instruction mix / register use / vmcnt waits of one kernel in a hipcc -S dump.
usage: isa_stats.py file.s <mangled-name-prefix>"""
import sys
from collections import Counter

path, prefix = sys.argv[1], sys.argv[2]
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith(prefix) and ':' in l)
end = next(i for i in range(start, len(lines)) if '.end_amdhsa_kernel' in lines[i])
body = lines[start:end]
for l in body:
    t = l.strip()
    if t.startswith(('.amdhsa_next_free_vgpr', '.amdhsa_next_free_sgpr', '.amdhsa_private_segment_fixed_size', '.amdhsa_accum_offset')):
        print(t)
ins = [l.strip() for l in body if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
c = Counter(i.split()[0] for i in ins)
print(len(ins), 'instructions;', c.most_common(30))
print('vmcnt waits:', Counter(l.split('vmcnt(')[1].split(')')[0] for l in ins if 'vmcnt(' in l))
print([l for l in ins if 'global_store' in l or 'global_atomic' in l or 'scratch_' in l][:20])
