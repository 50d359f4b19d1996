"""Memory/MFMA skeleton of one kernel's ISA: python tools/isa_skeleton.py file.s <mangled-name-substring>"""
import re, sys
lines = open(sys.argv[1]).read().splitlines()
start = next(i for i, l in enumerate(lines) if sys.argv[2] in l and l.rstrip().split(';')[0].strip().endswith(':'))
out, run = [], None
def flush():
    global run
    if run: out.append('%s x%d' % run)
    run = None
for l in lines[start + 1:]:
    t = l.strip()
    if t.startswith('s_endpgm'): break
    m = re.match(r'(global_load_lds_\w+|global_load_\w+|global_store_\w+|global_atomic_\w+|s_waitcnt|s_barrier|v_mfma_\w+|ds_read\w*|ds_write\w*|scratch_\w+|s_cbranch\w+|v_exp_f32)', t)
    if not m: continue
    k = m.group(1)
    if k == 's_waitcnt': k = t.split(';')[0].strip()
    if k.startswith('v_mfma'): k = 'mfma'
    if k.startswith('ds_read'): k = 'ds_read'
    if k.startswith('ds_write'): k = 'ds_write'
    if run and run[0] == k: run = (k, run[1] + 1)
    else:
        flush(); run = (k, 1)
flush()
print('\n'.join(out))
