"""Where the non-library kernels of a step sit: for every torch elementwise / runtime copy launch of the LAST step in a rocprofv3
kernel trace (rocpd .db), its duration and the library kernels launched just before and after.  python tools/prof_glue.py t_results.db"""
import re
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
rows = list(cur.execute('select name, start, end from kernels order by start'))
# the last optimizer launch but one delimits the last full step
opt = [i for i, r in enumerate(rows) if 'k_adamw' in r[0]]
lo, hi = (opt[-2] + 1, opt[-1] + 1) if len(opt) >= 2 else (0, len(rows))
step = rows[lo:hi]


def short(n):
    m = re.search(r'(k_\w+)(<[^>]*>)?', n)
    return (m.group(1) + (m.group(2) or '')) if m else re.sub(r'\s+', ' ', n)[:110]


tot = 0.0
for i, (n, s, e) in enumerate(step):
    if 'k_' in n and 'rocclr' not in n:
        continue
    tot += (e - s) / 1e3
    prev = next((short(step[j][0]) for j in range(i - 1, -1, -1) if 'k_' in step[j][0] and 'rocclr' not in step[j][0]), '-')
    nxt = next((short(step[j][0]) for j in range(i + 1, len(step)) if 'k_' in step[j][0] and 'rocclr' not in step[j][0]), '-')
    print('%8.1f us  %-110s  after %-28s before %s' % ((e - s) / 1e3, short(n), prev, nxt))
print('total %.1f us in %d launches of the step (%d kernels)' % (tot, sum(1 for r in step if not ('k_' in r[0] and 'rocclr' not in r[0])), len(step)))
