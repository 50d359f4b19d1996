"""Instruction mix of one kernel in a hipcc -S listing: python tools/isa_mix.py file.s name-substring [...]."""
import re
import sys
from collections import Counter

lines = open(sys.argv[1]).read().split('\n')
starts = [(i, l.split(':')[0]) for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l)]
for pat in sys.argv[2:]:
    for k, (i, name) in enumerate(starts):
        if pat in name:
            end = starts[k + 1][0] if k + 1 < len(starts) else len(lines)
            body = lines[i:end]
            c = Counter(m.group(1) for l in body for m in [re.match(r'^\s+([a-z][a-z_0-9]+)\s', l)] if m)
            top = sorted(c.items(), key=lambda kv: -kv[1])[:28]
            print(name[:70], 'total', sum(c.values()))
            print('   ', ', '.join('%s %d' % kv for kv in top))
