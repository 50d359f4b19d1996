"""Average rocprofv3 --pmc counter values per kernel from one or more rocpd sqlite .db files.
Usage: python tools/pmc_summary.py <filter-substring> a_results.db [b_results.db ...]"""
import collections
import re
import sqlite3
import sys


def collect(paths, needle):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in paths:
        cur = sqlite3.connect(p).cursor()
        for name, ctr, val in cur.execute('select kernel_name, counter_name, value from counters_collection'):
            if needle in name:
                m = re.search(r'(k_\w+)(<[^>]*>)?', name)
                out[(m.group(1) + (m.group(2) or '')) if m else name[:50]][ctr].append(val)
    return out


if __name__ == '__main__':
    res = collect(sys.argv[2:], sys.argv[1])
    for k in sorted(res):
        v = {c: sum(x) / len(x) for c, x in res[k].items()}
        print(k, ' (n=%d)' % max(len(x) for x in res[k].values()))
        if 'FETCH_SIZE' in v or 'WRITE_SIZE' in v:
            f, w = v.get('FETCH_SIZE', 0) * 1024, v.get('WRITE_SIZE', 0) * 1024
            print('   FETCH_SIZE %.1f MB raw (x2 = %.1f MB for 16 B/lane reads)  WRITE_SIZE %.1f MB' % (f / 1e6, 2 * f / 1e6, w / 1e6))
        for c in sorted(v):
            if c not in ('FETCH_SIZE', 'WRITE_SIZE'):
                print('   %-28s %.4g' % (c, v[c]))
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in v and 'SQ_BUSY_CYCLES' in v:
            print('   MFMA pipe busy %.0f %%' % (100 * v['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * v['SQ_BUSY_CYCLES'] / 32)))
