"""profiles/r02_f_pmc_wrb_fwd_C32.{txt,json} from the three rocprofv3 --pmc passes of tools/r02_final_profile.sh.
Usage: python tools/pmc_wrb_summary.py gpurun_out/r02g"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
txt = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'pmc_summary.py'), 'k_wrb_conv'] +
                     [os.path.join(src, d, 'p_results.db') for d in ('pmc_fetch', 'pmc_write', 'pmc_sq')],
                     capture_output=True, text=True, check=True).stdout
out = ['# rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ_* in separate runs, tools/r02_final_profile.sh) of tools/kb_wide.py at the bench shape',
       '# (B 64, C 32, H 65, T 1024, bf16 channels-last).  k_wrb_conv<32, D, MODE, SAVE>: MODE 0 = block forward (SAVE: hidden activation stored),',
       '# MODE 1 = data gradient.  Corrected traffic = FETCH_SIZE x 2 (gfx950 correction for 16 B/lane reads, MI355X_MICROARCH.md) + WRITE_SIZE.', '']
fetch, write, busy = {}, {}, {}
for b in re.split(r'\n(?=k_wrb_conv)', txt):
    m = re.match(r'k_wrb_convILi32ELi(\d)ELi(\d)ELb(\d)', b)
    if not m:
        continue
    d, mode, save = map(int, m.groups())
    f = float(re.search(r'FETCH_SIZE ([\d.]+) MB raw', b).group(1))
    w = float(re.search(r'WRITE_SIZE ([\d.]+) MB', b).group(1))
    out.append('k_wrb_conv<32, D=%d, MODE=%d, SAVE=%d>   FETCH_SIZE %.1f MB raw (x2 = %.1f MB)   WRITE_SIZE %.1f MB   corrected traffic %.1f MB'
               % (d, mode, save, f, 2 * f, w, 2 * f + w))
    out += ['    ' + l.strip() for l in b.split('\n')[2:] if l.strip()]
    if mode == 0 and save == 1:
        fetch[d], write[d] = f * 1e6, w * 1e6
        busy[d] = int(re.search(r'MFMA pipe busy (\d+)', b).group(1))
open(os.path.join(ROOT, 'profiles', 'r02_f_pmc_wrb_fwd_C32.txt'), 'w').write('\n'.join(out) + '\n')
tr = {d: 2 * fetch[d] + write[d] for d in fetch}
js = {'kernel': 'k_wrb_conv<32,D,0,true> (fused ResidualConv2dBlock forward, bf16 channels-last, hidden activation saved)',
      'shape': {'B': 64, 'C': 32, 'H': 65, 'T': 1024},
      'fetch_size_raw_bytes': {str(d): fetch[d] for d in sorted(fetch)},
      'write_size_bytes': {str(d): write[d] for d in sorted(write)},
      'traffic_bytes_corrected_per_dilation': {str(d): tr[d] for d in sorted(tr)},
      'traffic_bytes_corrected': sum(tr.values()) / len(tr),
      'algorithmic_bytes': {'x_plus_y': 545259520, 'x_plus_y_plus_saved_h1': 817889280},
      'note': 'FETCH_SIZE x2 (gfx950 correction, MI355X_MICROARCH.md HBM section) + WRITE_SIZE, mean over dilations 1,2,3; source profiles/r02_f_pmc_wrb_fwd_C32.txt',
      'mfma_pipe_busy_pct': {str(d): busy[d] for d in sorted(busy)}}
json.dump(js, open(os.path.join(ROOT, 'profiles', 'r02_f_pmc_wrb_fwd_C32.json'), 'w'), indent=1)
print(json.dumps(js['traffic_bytes_corrected_per_dilation']), js['mfma_pipe_busy_pct'])
