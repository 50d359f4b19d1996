"""Register-limited occupancy of every kernel in hipcc -S listings: python tools/occupancy.py a.s [b.s ...]
(waves/SIMD = min(8, 512 // alloc) with alloc = VGPR + AGPR rounded up to 8, MI355X_MICROARCH.md register-file table)."""
import re
import subprocess
import sys


def demangle(n):
    try:
        r = subprocess.run(['c++filt', n], capture_output=True, text=True).stdout.strip()
    except FileNotFoundError:
        r = n
    r = re.sub(r'\(anonymous namespace\)::', '', r)
    return re.sub(r'\(.*', '', r).replace('void ', '')


for f in sys.argv[1:]:
    cur = {}
    for line in open(f):
        line = line.strip()
        m = re.match(r'-?\s*\.(name|sgpr_count|vgpr_count|agpr_count|group_segment_fixed_size):\s+(\S+)', line)
        if not m:
            continue
        k, v = m.groups()
        if k in cur:                       # next kernel's block started
            if 'name' in cur and 'vgpr_count' in cur:
                tot = int(cur['vgpr_count']) + int(cur.get('agpr_count', 0))
                alloc = (tot + 7) // 8 * 8
                print('%-56s vgpr %3d  waves/SIMD %d  sgpr %s' % (demangle(cur['name'])[:56], tot, min(8, 512 // max(alloc, 8)), cur.get('sgpr_count')))
            cur = {}
        cur[k] = v
    if 'name' in cur and 'vgpr_count' in cur:
        tot = int(cur['vgpr_count']) + int(cur.get('agpr_count', 0))
        alloc = (tot + 7) // 8 * 8
        print('%-56s vgpr %3d  waves/SIMD %d  sgpr %s' % (demangle(cur['name'])[:56], tot, min(8, 512 // max(alloc, 8)), cur.get('sgpr_count')))
