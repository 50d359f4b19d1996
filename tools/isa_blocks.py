"""Per-basic-block instruction mix of one kernel in a hipcc -S listing: python tools/isa_blocks.py file.s name-substring [min instrs]."""
import re
import sys

lines = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2]
mininstr = int(sys.argv[3]) if len(sys.argv) > 3 else 30
start = [i for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l) and pat in l][0]
end = [i for i, l in enumerate(lines[start:]) if 's_endpgm' in l][0] + start
cur, cnt, order = 'entry', {}, ['entry']
cnt[cur] = dict(n=0, mfma=0, valu=0, exp=0, ds=0, vmem=0, salu=0, wait=0, line=start)
for i, l in enumerate(lines[start:end]):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        cur = m.group(1)
        order.append(cur)
        cnt[cur] = dict(n=0, mfma=0, valu=0, exp=0, ds=0, vmem=0, salu=0, wait=0, line=start + i)
    elif re.match(r'^\s+[a-z]', l):
        op = l.split()[0]
        c = cnt[cur]
        c['n'] += 1
        if 'mfma' in op: c['mfma'] += 1
        elif op.startswith('v_exp'): c['exp'] += 1; c['valu'] += 1
        elif op.startswith('v_'): c['valu'] += 1
        elif op.startswith('ds_'): c['ds'] += 1
        elif op.startswith(('global_', 'buffer_', 'scratch_')): c['vmem'] += 1
        elif op.startswith(('s_waitcnt', 's_nop')): c['wait'] += 1
        elif op.startswith('s_'): c['salu'] += 1
for b in order:
    if cnt[b]['n'] >= mininstr:
        print(b, cnt[b])
