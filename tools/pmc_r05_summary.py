"""profiles/r05_pmc_*.txt and profiles/r05_pmc_wrb_bwd_C32.json from the summaries tools/r05_profile.sh / tools/r05_pmc_x3n.sh leave in
gpurun_out/pmc_r05_*/summary.txt (rocprofv3 --pmc passes: SQ x2, FETCH_SIZE, WRITE_SIZE, separate runs).

Per kernel: HBM traffic = FETCH_SIZE x 2 (gfx950 correction for 16 B/lane reads, MI355X_MICROARCH.md) + WRITE_SIZE against the algorithmic
bytes of the launch; duration = SQ_BUSY_CYCLES / 32 shader engines (cycles); vector-ALU busy = SQ_ACTIVE_INST_VALU x 4 (quad-cycles ->
cycles) / 1024 SIMDs / duration; matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 / duration; parked = SQ_WAIT_ANY / SQ_WAVE_CYCLES."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H = {32: 65, 16: 133, 8: 269, 4: 540}


def parse(tag):
    cur, d = None, {}
    for line in open(os.path.join(ROOT, 'gpurun_out', 'pmc_' + tag, 'summary.txt')):
        if line.startswith('k_'):
            cur = line.split('  (n=')[0].strip()
            d[cur] = {}
        elif cur:
            m = re.match(r'\s+(SQ_\w+)\s+([\d.e+]+)', line)
            if m:
                d[cur][m.group(1)] = float(m.group(2))
            m = re.search(r'FETCH_SIZE ([\d.]+) MB raw.*WRITE_SIZE ([\d.]+)', line)
            if m:
                d[cur]['F'], d[cur]['W'] = float(m.group(1)), float(m.group(2))
    return d


def short(name):
    m = re.match(r'k_(\w+?)ILi(\d+)ELi(\d)ELi(\d)ELb(\d)ELi(\d)', name)
    if m:
        return 'k_%s<%s,%s,%s,%s,%s>' % m.groups()
    return name


def rows(d, alg_of, min_cycles=3e5):
    out, js = [], {}
    for k in sorted(d, key=short):
        v = d[k]
        if v.get('SQ_BUSY_CYCLES', 0) < min_cycles or 'F' not in v:
            continue
        dur = v['SQ_BUSY_CYCLES'] / 32
        traffic = 2 * v['F'] + v['W']
        alg, what = alg_of(short(k))
        wc = v.get('SQ_WAVE_CYCLES', 0) or 1
        valu, mfma = 4 * v.get('SQ_ACTIVE_INST_VALU', 0) / 1024 / dur, v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024 / dur
        out.append('%-36s traffic %7.1f MB  alg %7.1f (%s) %s | dur %6.0fk cyc | VALU %3.0f%% MFMA %3.0f%% | parked %2.0f%% | VALU insts %5.1fM LDS insts %5.1fM | waves %d'
                   % (short(k), traffic, alg or 0, what, ('%.2fx' % (traffic / alg)) if alg else ' - ', dur / 1e3, 100 * valu, 100 * mfma,
                      100 * v.get('SQ_WAIT_ANY', 0) / wc, v.get('SQ_INSTS_VALU', 0) / 1e6, v.get('SQ_INSTS_LDS', 0) / 1e6, int(v.get('SQ_WAVES', 0))))
        js[short(k)] = dict(traffic_mb=traffic, algorithmic_mb=alg, duration_kcycles=dur / 1e3, valu_busy=valu, mfma_busy=mfma)
    return out, js


def bf16_alg(B=64, T=1024):
    def f(name):
        m = re.search(r'<(\d+)', name)
        C = int(m.group(1)) if m else 32
        t = 2.0 * B * C * H.get(C, 65) * T / 1e6
        if 'bwd_a' in name: return 3 * t, 'h1, dy, dA1'
        if 'bwd_fused' in name: return 4 * t, 'h1, dy, x, dx'
        if 'dxw' in name: return 4 * t, 'dA1, dy, x, dx'
        if 'conv' in name: return 3 * t, 'x, y, h1'
        return None, ''
    return f


def x3n_alg(name):
    m = re.search(r'<(\d+)', name)
    C = int(m.group(1))
    return 2.0 * 96 * C * H[C] * 1024 * 4 / 1e6, 'x, y at 4 bytes per element, 96 chunks'


def main():
    head = ['# rocprofv3 --pmc, bf16 channels-last tensors at the bench shapes (B 64, T 1024; H = 65 / 269 / 540 at C = 32 / 8 / 4), MI355X, round 5.',
            '# traffic = FETCH_SIZE x 2 + WRITE_SIZE (MB); alg = bytes the launch must move once; dur = SQ_BUSY_CYCLES / 32 (k cycles);',
            '# VALU / MFMA = busy share of the 1024 SIMDs / matrix pipes over the launch; parked = share of the waves\' life in s_waitcnt / barriers.', '']
    d32 = parse('r05_C32')
    lines, j32 = rows(d32, bf16_alg())
    open(os.path.join(ROOT, 'profiles', 'r05_pmc_bwd_C32.txt'), 'w').write(
        '\n'.join(['# tt_wide_rb_bwd at C = 32, default dispatch: k_wrb_bwd_a<32> + k_wrb_dxw<32,D,8,32> (halo-free x tile, round 5) + k_wrb_reduce<32>'] + head + lines) + '\n')
    per_d = {}
    for dd in (1, 2, 3):
        dxw = [k for k in j32 if k.startswith('k_wrb_dxw<32, %d,' % dd)]
        assert len(dxw) == 1, sorted(j32)
        red = d32.get('k_wrb_reduce<32>', {})
        per_d[dd] = (j32['k_wrb_bwd_a<32>']['traffic_mb'] + j32[dxw[0]]['traffic_mb'] + 2 * red.get('F', 0) + red.get('W', 0)) * 1e6
    mean = sum(per_d.values()) / 3
    json.dump(dict(call='tt_wide_rb_bwd at C = 32 (k_wrb_bwd_a<32> + k_wrb_dxw<32,D,8,32> + k_wrb_reduce<32>), round 5',
                   shape=dict(B=64, C=32, H=65, T=1024), traffic_bytes_corrected_per_dilation={str(k): v for k, v in per_d.items()},
                   traffic_bytes_corrected=mean, algorithmic_bytes=dict(dy_x_dx=817889280), kernels=j32,
                   summary='traffic %.2f GB per call against 0.818 GB algorithmic (%.2fx): h1 read, dA1 written and read once, dy read twice; the '
                           'halo-free x tile took the merged kernel\'s fetch from 1.01-1.06 GB to 0.86-0.93 GB' % (mean / 1e9, mean / 817889280),
                   note='FETCH_SIZE x2 (gfx950 correction, MI355X_MICROARCH.md HBM section) + WRITE_SIZE; source profiles/r05_pmc_bwd_C32.txt'),
              open(os.path.join(ROOT, 'profiles', 'r05_pmc_wrb_bwd_C32.json'), 'w'), indent=1)
    lines, _ = rows({k: v for k, v in parse('r05_narrow').items() if 'bwd_fused' in k or 'reduce' in k}, bf16_alg())
    open(os.path.join(ROOT, 'profiles', 'r05_pmc_bwd_narrow.txt'), 'w').write(
        '\n'.join(['# tt_wide_rb_bwd at C = 8, 4: k_nrb_bwd_fused (the whole backward of a narrow block in one pass; round-5 cut: halo-free x tile, deferred x wait, precomputed phase-1 indices) + k_nrb_reduce'] + head + lines) + '\n')
    if os.path.exists(os.path.join(ROOT, 'gpurun_out', 'pmc_r05_x3n', 'summary.txt')):
        lines, _ = rows(parse('r05_x3n'), x3n_alg)
        open(os.path.join(ROOT, 'profiles', 'r05_pmc_x3n.txt'), 'w').write(
            '\n'.join(['# tt_x3n_rb_fwd (narrow split-operand blocks of the no-grad fp32 forward) at the inference shape of BASELINE configs[1] (96 chunks; tools/kb_x3n.py): k_x3n_conv<C, D, planar in, planar out>',
                       '# measured on the first 16x16x16 cut (profiles/r05_x3n_shapes.txt for the shapes tried since); fp32 planar or split tensors, 4 bytes per element either way'] + head[1:] + lines) + '\n')
    print(open(os.path.join(ROOT, 'profiles', 'r05_pmc_bwd_C32.txt')).read())
    print(json.load(open(os.path.join(ROOT, 'profiles', 'r05_pmc_wrb_bwd_C32.json')))['summary'])


if __name__ == '__main__':
    main()
