"""Launch geometry of the kernels in a rocprofv3 kernel trace (rocpd .db): workgroups per launch against the CU count, LDS and registers.
python tools/prof_geometry.py t_results.db"""
import re
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
want = [c for c in ('name', 'grid_x', 'grid_size', 'workgroup_x', 'workgroup_size', 'lds_size', 'lds_block_size', 'arch_vgpr_count', 'accum_vgpr_count', 'sgpr_count', 'scratch_size') if c in cols]
if 'name' not in want:
    print('columns:', cols); sys.exit(0)
rows = list(cur.execute('select %s, count(*), avg(end-start)/1e3 from kernels group by %s order by sum(end-start) desc limit 60' % (', '.join(want), ', '.join(want))))
print(' | '.join(want + ['launches', 'avg us']))
for r in rows:
    name = r[0]
    m = re.search(r'(k_\w+)(<[^>]*>)?', name)
    short = (m.group(1) + (m.group(2) or '')) if m else name[:40]
    print(short[:42].ljust(42), ' | '.join(str(v) for v in r[1:-1]), '| %.1f' % r[-1])
