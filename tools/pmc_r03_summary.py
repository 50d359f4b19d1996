"""profiles/r03_pmc_bwd_C{4,8,16,32}.txt, profiles/r03_pmc_fused.txt and profiles/r03_pmc_wrb_bwd_C32.json from the summaries
tools/r03_pmc_bwd.sh leaves in gpurun_out/pmc_r03_*/summary.txt (rocprofv3 --pmc passes: SQ x2, FETCH_SIZE, WRITE_SIZE, separate runs).

Per kernel: HBM traffic = FETCH_SIZE x 2 (gfx950 correction for 16 B/lane reads, MI355X_MICROARCH.md) + WRITE_SIZE against the
algorithmic bytes of the launch; duration = SQ_BUSY_CYCLES / 32 shader engines (cycles); vector-ALU busy = SQ_ACTIVE_INST_VALU x 4
(quad-cycles -> cycles) / 1024 SIMDs / duration; matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 / duration; a wave's life split
into parked (s_waitcnt / barrier), issue-stalled and issuing."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = {32: 65, 16: 133, 8: 269, 4: 540}
B, T = 64, 1024


def parse(path):
    out, cur = {}, None
    for line in open(path):
        if line.startswith('k_'):
            cur = line.split('  (n=')[0].strip()
            out[cur] = {}
        elif cur and line.startswith('   FETCH_SIZE'):
            m = re.search(r'FETCH_SIZE ([\d.]+) MB raw.*WRITE_SIZE ([\d.]+) MB', line)
            out[cur]['FETCH_MB'], out[cur]['WRITE_MB'] = float(m.group(1)), float(m.group(2))
        elif cur and line.startswith('   SQ_'):
            k, v = line.split()
            out[cur][k] = float(v)
    return out


def pretty(name):
    m = re.match(r'k_(\w+?)ILi(\d+)ELi(\d)ELi(\d)ELb(\d)', name)
    if m:
        return 'k_%s<%s,%s,%s,%s>' % m.groups()
    m = re.match(r'k_wrb_bwd_fusedILi(\d+)ELi(\d)ELi(\d)ELi(\d+)', name)
    if m:
        return 'k_wrb_bwd_fused<%s,%s,%s,%s>' % m.groups()
    m = re.match(r'k_(\w+?)ILi(\d+)ELi(\d)ELi(\d)ELb(\d)ELi(\d)', name)
    if m:
        return 'k_%s<%s,%s,%s,%s>' % m.groups()[:5]
    return re.sub(r'E[vP].*', '', name)


def algorithmic_mb(name, C):
    """bytes a launch must move once, MB (bf16 tensors of B x C x H x T elements; the strided layers' two sides have equal bytes)."""
    t = 2.0 * B * C * SHAPES[C] * T / 1e6
    p = pretty(name)
    if 'conv<' in p and p.endswith(',0,1>'): return 3 * t, 'x, y, h1'
    if 'conv<' in p and p.endswith(',0,0>'): return 2 * t, 'x, y'
    if 'conv<' in p: return 3 * t, 'dA1, dy, dx'
    if 'bwd_a' in p: return 3 * t, 'h1, dy, dA1'
    if 'bwd_fused' in p and 'nrb' in p: return 4 * t, 'h1, dy, x, dx'
    if 'bwd_fused' in p: return 3 * t, 'x, dy, dx'
    if 'dxw' in p: return 4 * t, 'dA1, dy, x, dx'
    if 'wgrad' in p: return 2 * t, 'x, dA1'
    if p.startswith(('k_s4', 'k_p2')): return 3 * t, 'dy, y (gate), dx'
    if p.startswith('k_w4<') and p.rstrip('>').endswith('true'): return 4 * t, 'x, dy, y (gate), dx'
    if p.startswith('k_w4<'): return 3 * t, 'x, dy, y (gate)'
    return None, ''


def report(tag, C_list, title):
    lines = ['# ' + title,
             '# rocprofv3 --pmc, bench shapes (B 64, T 1024; H = 65 / 133 / 269 / 540 at C = 32 / 16 / 8 / 4), bf16 channels-last tensors, MI355X.',
             '# traffic = FETCH_SIZE x 2 + WRITE_SIZE (MB); alg = bytes the launch must move once; dur = SQ_BUSY_CYCLES / 32 (k cycles);',
             '# VALU / MFMA = busy share of the 1024 SIMDs / matrix pipes over the launch; parked / stalled / issuing = split of the waves\' life.', '']
    js = {}
    path = os.path.join(ROOT, 'gpurun_out', 'pmc_' + tag, 'summary.txt')
    data = parse(path)
    for name in sorted(data, key=pretty):
        v = data[name]
        if 'SQ_BUSY_CYCLES' not in v or v['SQ_BUSY_CYCLES'] < 3e5:
            continue
        p = pretty(name)
        m = re.search(r'<(\d+)', p)
        C = int(m.group(1)) if m else C_list[0]
        if C not in SHAPES:
            continue
        dur = v['SQ_BUSY_CYCLES'] / 32
        traffic = 2 * v.get('FETCH_MB', 0) + v.get('WRITE_MB', 0)
        alg, what = algorithmic_mb(name, C)
        valu = 4 * v.get('SQ_ACTIVE_INST_VALU', 0) / 1024 / dur
        mfma = v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024 / dur
        wc = v.get('SQ_WAVE_CYCLES', 0) or 1
        parked, stalled, issuing = v.get('SQ_WAIT_ANY', 0) / wc, v.get('SQ_WAIT_INST_ANY', 0) / wc, v.get('SQ_ACTIVE_INST_ANY', 0) / wc
        conf = v.get('SQ_LDS_BANK_CONFLICT', 0) / max(v.get('SQ_LDS_IDX_ACTIVE', 0), 1)
        ratio = '%.2fx' % (traffic / alg) if alg else ' -  '
        lines.append('%-34s traffic %7.1f MB  alg %7.1f (%s) %s | dur %6.0fk cyc | VALU %3.0f%% MFMA %3.0f%% | parked %2.0f%% stalled %2.0f%% issuing %2.0f%% | '
                     'VALU insts %5.1fM | LDS conflict share %2.0f%%'
                     % (p, traffic, alg or 0, what, ratio, dur / 1e3, 100 * valu, 100 * mfma, 100 * parked, 100 * stalled, 100 * issuing,
                        v.get('SQ_INSTS_VALU', 0) / 1e6, 100 * conf))
        js[p] = dict(traffic_mb=traffic, algorithmic_mb=alg, traffic_ratio=(traffic / alg) if alg else None, duration_kcycles=dur / 1e3,
                     valu_busy=valu, mfma_busy=mfma, parked=parked, stalled=stalled, issuing=issuing)
    return lines, js


def main():
    allj = {}
    for C in (32, 16, 8, 4):
        lines, js = report('r03_C%d' % C, [C], 'Residual-block forward / backward kernels and the strided-layer backward at C = %d' % C)
        open(os.path.join(ROOT, 'profiles', 'r03_pmc_bwd_C%d.txt' % C), 'w').write('\n'.join(lines) + '\n')
        allj[C] = js
    lines, js = report('r03_fused', [32], 'One-pass residual backward with recomputed hidden activation (csrc/conv_level_bf16.hip, opt-in)')
    open(os.path.join(ROOT, 'profiles', 'r03_pmc_fused.txt'), 'w').write('\n'.join(lines) + '\n')
    cq = parse(os.path.join(ROOT, 'gpurun_out', 'pmc_r03_cqt', 'summary.txt'))
    lines = ['# CQT kernels, 64 clips x 3 s (tools/kbench.py cqt under rocprofv3 --pmc; forward = k_fft675_rows + k_fft49_cols<0> + k_band_fwd,',
             '# inverse = k_band_inv + k_spec_gather + k_fft675_rows + k_fft49_cols<1> + k_scale_by_max).  Algorithmic bytes of either direction: 300.0 MB',
             '# (64 x 4,688,280 B).  traffic = FETCH_SIZE x 2 + WRITE_SIZE; dur = SQ_BUSY_CYCLES / 32; VALU = SQ_ACTIVE_INST_VALU x 4 / 1024 / dur.', '']
    tot = {}
    for name in sorted(cq):
        v = cq[name]
        if 'SQ_BUSY_CYCLES' not in v:
            continue
        dur = v['SQ_BUSY_CYCLES'] / 32
        wc = v.get('SQ_WAVE_CYCLES', 0) or 1
        lines.append('%-20s traffic %6.1f MB (fetch x2 %5.1f + write %5.1f) | dur %6.1fk cyc | VALU %3.0f%% | parked %2.0f%% stalled %2.0f%% issuing %2.0f%% | VALU insts %5.2fM'
                     % (name.split('  ')[0], 2 * v.get('FETCH_MB', 0) + v.get('WRITE_MB', 0), 2 * v.get('FETCH_MB', 0), v.get('WRITE_MB', 0), dur / 1e3,
                        100 * 4 * v.get('SQ_ACTIVE_INST_VALU', 0) / 1024 / dur, 100 * v.get('SQ_WAIT_ANY', 0) / wc, 100 * v.get('SQ_WAIT_INST_ANY', 0) / wc,
                        100 * v.get('SQ_ACTIVE_INST_ANY', 0) / wc, v.get('SQ_INSTS_VALU', 0) / 1e6))
        tot[name.split('  ')[0]] = dur
    open(os.path.join(ROOT, 'profiles', 'r03_pmc_cqt.txt'), 'w').write('\n'.join(lines) + '\n')
    # boundary convolutions and the strided layers' forward kernels (gpurun_out/pmc_r03_misc)
    mi = parse(os.path.join(ROOT, 'gpurun_out', 'pmc_r03_misc', 'summary.txt'))
    px = B * 540 * T
    edge_alg = {'k_cin_fwd': (16, '8 B in + 8 B out per pixel'), 'k_cin_bwd<true': (32, 'x, y, dy, dx'), 'k_cin_bwd<false': (24, 'x, y, dy'),
                'k_cout_fwd': (16, '8 B in + 8 B out'), 'k_cout_bwd': (24, 'x, dy, dx')}
    lines = ['# Boundary 3x3 convolutions (H = 540, B 64, T 1024) and the strided / transposed layers FORWARD at every width (tools/kb_level.py, KB_WHAT=stridefwd,edge)',
             '# under rocprofv3 --pmc.  Same columns as profiles/r03_pmc_bwd_C*.txt; alg for the strided layers = input + output once (equal bytes on both sides).', '']
    for name in sorted(mi):
        v = mi[name]
        if 'SQ_BUSY_CYCLES' not in v or v['SQ_BUSY_CYCLES'] < 3e5:
            continue
        short = name.split('  ')[0]
        alg = None
        for key, (bpp, what) in edge_alg.items():
            if short.startswith(key):
                alg = px * bpp / 1e6
        m = re.match(r'k_(s4n?|p2n?)<(\d+)', short)
        if m:
            C = int(m.group(2))
            alg = 2 * 2.0 * B * C * SHAPES[C] * T / 1e6
        dur = v['SQ_BUSY_CYCLES'] / 32
        wc = v.get('SQ_WAVE_CYCLES', 0) or 1
        traffic = 2 * v.get('FETCH_MB', 0) + v.get('WRITE_MB', 0)
        lines.append('%-28s traffic %7.1f MB  alg %7.1f %s | dur %6.0fk cyc | VALU %3.0f%% MFMA %3.0f%% | parked %2.0f%% stalled %2.0f%% issuing %2.0f%% | VALU insts %5.1fM | LDS insts %5.1fM'
                     % (short, traffic, alg or 0, ('%.2fx' % (traffic / alg)) if alg else ' -  ', dur / 1e3, 100 * 4 * v.get('SQ_ACTIVE_INST_VALU', 0) / 1024 / dur,
                        100 * v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024 / dur, 100 * v.get('SQ_WAIT_ANY', 0) / wc, 100 * v.get('SQ_WAIT_INST_ANY', 0) / wc,
                        100 * v.get('SQ_ACTIVE_INST_ANY', 0) / wc, v.get('SQ_INSTS_VALU', 0) / 1e6, v.get('SQ_INSTS_LDS', 0) / 1e6))
    open(os.path.join(ROOT, 'profiles', 'r03_pmc_edge_stridefwd.txt'), 'w').write('\n'.join(lines) + '\n')
    # the bench line's roofline call: tt_wide_rb_bwd at C = 32 = k_wrb_bwd_a<32> + k_wrb_dxw<32,D,8,32> + k_wrb_reduce<32>, mean over D
    j32 = allj[32]
    per_d = {}
    for d in (1, 2, 3):
        dxw = [k for k in j32 if k.startswith('k_wrb_dxw<32, %d,' % d)]
        assert len(dxw) == 1, (d, sorted(j32))
        parts = ['k_wrb_bwd_a<32>', dxw[0], 'k_wrb_reduce<32>']
        missing = [q for q in parts if q not in j32]
        assert not missing, (missing, sorted(j32))
        per_d[d] = sum(j32[q]['traffic_mb'] for q in parts) * 1e6
    kern = {k: v for k, v in j32.items() if k.startswith(('k_wrb_bwd_a', 'k_wrb_dxw', 'k_wrb_reduce'))}
    big = [v for v in kern.values() if v['duration_kcycles'] > 100]
    out = dict(call='tt_wide_rb_bwd at C = 32 (k_wrb_bwd_a<32> + k_wrb_dxw<32,D,TH,32> + k_wrb_reduce<32>)',
               shape=dict(B=B, C=32, H=65, T=T), traffic_bytes_corrected_per_dilation={str(d): v for d, v in per_d.items()},
               traffic_bytes_corrected=sum(per_d.values()) / 3, algorithmic_bytes=dict(dy_x_dx=817889280),
               kernels=kern,
               summary='traffic %.2f GB per call against 0.818 GB algorithmic (%.2fx): h1 read, dA1 written and read once, dy read twice; '
                       'vector ALU %.0f-%.0f %% busy, matrix pipe %.0f-%.0f %%'
                       % (sum(per_d.values()) / 3 / 1e9, sum(per_d.values()) / 3 / 817889280,
                          100 * min(v['valu_busy'] for v in big), 100 * max(v['valu_busy'] for v in big),
                          100 * min(v['mfma_busy'] for v in big), 100 * max(v['mfma_busy'] for v in big)),
               note='FETCH_SIZE x2 (gfx950 correction, MI355X_MICROARCH.md HBM section) + WRITE_SIZE; source profiles/r03_pmc_bwd_C32.txt')
    json.dump(out, open(os.path.join(ROOT, 'profiles', 'r03_pmc_wrb_bwd_C32.json'), 'w'), indent=1)
    print(out['summary'])


if __name__ == '__main__':
    main()
