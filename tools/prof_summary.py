"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd sqlite .db) as text: per-kernel count / total / average."""
import collections
import re
import sqlite3
import sys


def main(db_path, steps):
    cur = sqlite3.connect(db_path).cursor()
    rows = list(cur.execute('select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3 from kernels group by name order by 3 desc'))
    tot = sum(r[2] for r in rows)
    print('# rocprofv3 --kernel-trace --stats summary of %s' % db_path)
    print('# total kernel time %.1f ms over %d steps (%.1f ms/step)' % (tot, steps, tot / steps))
    fam = collections.defaultdict(lambda: [0.0, 0])
    for name, n, ms, _ in rows:
        m = re.search(r'(k_\w+)(<[^>]*>)?', name)
        key = (m.group(1) + (m.group(2) or '')) if m else re.sub(r'\s+', ' ', name)[:200]     # (torch's own kernels: enough of the name to tell them apart)
        fam[key][0] += ms
        fam[key][1] += n
    print('\n## by kernel (ms per step, %% of kernel time, launches per step, avg us)')
    for k, (ms, n) in sorted(fam.items(), key=lambda x: -x[1][0]):
        print('%9.2f ms/step %5.1f%%  n/step=%6.1f  avg=%9.1f us  %s' % (ms / steps, 100 * ms / tot, n / steps, 1e3 * ms / n, k))


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1)
