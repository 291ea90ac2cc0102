#!/usr/bin/env python3
"""Static instruction mix and register use of the kernels in a hipcc -S listing (offline: no GPU needed).
usage: isa_count.py file.s [kernel-name-substring ...]"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read().split('\n')
want = sys.argv[2:]
i = 0
while i < len(s):
    m = re.match(r'^(_Z\w+):\s*;\s*@', s[i])
    if not m:
        i += 1
        continue
    name = m.group(1)
    j = i + 1
    ins = []
    while j < len(s) and not s[j].startswith('.Lfunc_end'):
        l = s[j].strip()
        if l and not l.startswith(('.', ';')) and not l.endswith(':'):
            ins.append(l.split()[0])
        j += 1
    if not want or any(w in name for w in want):
        c = Counter(x.split('_')[0] for x in ins)
        mem = Counter(x for x in ins if x.startswith(('s_load', 'global_', 'buffer_', 'ds_', 'scratch_', 'flat_')))
        print(name, 'instructions', len(ins), dict(c))
        print('   mem', dict(mem))
    i = j
txt = '\n'.join(s)
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', txt, re.S):
    if want and not any(w in m.group(1) for w in want):
        continue
    d = {k: re.search(r'\.amdhsa_%s (\d+)' % k, m.group(2)).group(1) for k in ('next_free_vgpr', 'next_free_sgpr', 'private_segment_fixed_size', 'group_segment_fixed_size')}
    print(m.group(1), d)
