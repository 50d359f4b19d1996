"""Per-kernel register / LDS / spill table of one .hip source (hipcc -Rpass-analysis=kernel-resource-usage).
Usage: python tools/resource_usage.py timbre-trap_amd/csrc/conv_mfma.hip [name-filter]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
out = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-I' + os.path.join(ROOT, 'include'),
                      '-c', src, '-o', '/dev/null', '-Rpass-analysis=kernel-resource-usage'], capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r'Function Name: (\S+)', line)
    if m:
        cur = {'name': subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
        continue
    m = re.search(r'remark:\s+([A-Za-z /\[\]]+): (\d+)', line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
for r in rows:
    n = re.sub(r'\(anonymous namespace\)::', '', r['name'])
    n = re.sub(r'\(.*', '', n).replace('void ', '')
    if flt in n:
        print('%-62s vgpr %3d agpr %3d spill %3d occ %d lds %6d' % (n[:62], r.get('VGPRs', -1), r.get('AGPRs', -1), r.get('VGPRs Spill', -1),
                                                                 r.get('Occupancy [waves/SIMD]', -1), r.get('LDS Size [bytes/block]', -1)))
