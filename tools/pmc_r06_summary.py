"""profiles/r06_pmc_bwds_C16.{txt,json} and profiles/r06_pmc_skip.txt from the summaries tools/r06_profile.sh leaves in gpurun_out/pmc_r06_*/summary.txt
(rocprofv3 --pmc passes: SQ x2, FETCH_SIZE, WRITE_SIZE, separate runs).  Definitions as tools/pmc_r05_summary.py."""
import json
import os
import re

from pmc_r05_summary import ROOT, parse, rows, short

HEAD = ['# rocprofv3 --pmc at the bench shapes (B 64, T 1024), bf16 channels-last tensors, MI355X, round 6.',
        '# traffic = FETCH_SIZE x 2 + WRITE_SIZE (MB); alg = bytes the launch must move once; dur = SQ_BUSY_CYCLES / 32 (k cycles);',
        '# VALU / MFMA = busy share of the 1024 SIMDs / matrix pipes over the launch; parked = share of the waves\' life in s_waitcnt / barriers.', '']
HS = {64: 31, 32: 65, 16: 133, 8: 269, 4: 540}


def c16_alg(name):
    t = 2.0 * 64 * 16 * 133 * 1024 / 1e6
    if 'bwds' in name: return 4 * t, 'h1, dy, x, dx'
    return None, ''


def main():
    d16 = parse('r06_C16')
    lines, j16 = rows(d16, c16_alg)
    conf = {k: (v.get('SQ_LDS_BANK_CONFLICT', 0) / max(v.get('SQ_LDS_IDX_ACTIVE', 0), 1)) for k, v in d16.items() if 'bwds' in k}
    extra = ['', '# LDS bank-conflict cycles / LDS active cycles: ' + ', '.join('%s %.0f %%' % (short(k), 100 * c) for k, c in sorted(conf.items()))]
    open(os.path.join(ROOT, 'profiles', 'r06_pmc_bwds_C16.txt'), 'w').write(
        '\n'.join(['# tt_wide_rb_bwd at C = 16 (H = 133): k_wrb_bwds<16,D,8,32,4,GOUT> (one-pass strip backward, the x rows in an LDS image of their own since round 5) + k_wrb_reduce<16>'] + HEAD + lines + extra) + '\n')
    strips = [v['traffic_mb'] for k, v in j16.items() if 'bwds' in k]
    red = d16.get('k_wrb_reduce<16>', {})
    mean = (sum(strips) / len(strips) + 2 * red.get('F', 0) + red.get('W', 0)) * 1e6
    alg = 3 * 2.0 * 64 * 16 * 133 * 1024
    json.dump(dict(call='tt_wide_rb_bwd at C = 16 (k_wrb_bwds<16,D,8,32> + k_wrb_reduce<16>), round 6', shape=dict(B=64, C=16, H=133, T=1024),
                   traffic_bytes_corrected=mean, algorithmic_bytes=dict(dy_x_dx=alg), kernels=j16,
                   summary='traffic %.2f GB per call against %.3f GB algorithmic (%.2fx; %.2fx of the four tensors the kernel moves)' % (mean / 1e9, alg / 1e9, mean / alg, mean / (alg * 4 / 3)),
                   note='FETCH_SIZE x2 (gfx950 correction, MI355X_MICROARCH.md HBM section) + WRITE_SIZE, mean of the three dilations; source profiles/r06_pmc_bwds_C16.txt'),
              open(os.path.join(ROOT, 'profiles', 'r06_pmc_bwds_C16.json'), 'w'), indent=1)

    ds = parse('r06_skip')

    def skip_alg(name):
        # tools/kb_skip.py runs every level: the summary averages over C = 4 .. 64 (all levels hold ~283 MB per 64-clip tensor, 260 MB at C = 64)
        unit = sum(2.0 * 64 * C * H * 1024 / 1e6 for C, H in HS.items()) / len(HS)
        if 'skip_join_fwd' in name: return 5 * unit, '2 y + e in, 2 out; mean over the five levels'
        if 'skip_join_bwd' in name:
            return (5 if name.rstrip('>').endswith('true') else 4) * unit, '2 g + e (+ de) in, de out; mean over the five levels'
        m = re.match(r'k_(w|n)rb_conv<(\d+),3,([02]),1', name.replace(' ', ''))
        if m:
            C = int(m.group(2)); t = 2.0 * 128 * C * HS[C] * 1024 / 1e6
            return (3 * t + (t / 2 if m.group(3) == '2' else 0)), 'x, y, h1 over 2 B clips' + (' + the embedding once' if m.group(3) == '2' else '')
        return None, ''
    lines, _ = rows(ds, skip_alg, min_cycles=1e5)
    open(os.path.join(ROOT, 'profiles', 'r06_pmc_skip.txt'), 'w').write(
        '\n'.join(['# the skip-join kernels of round 6 (tools/kb_skip.py): k_skip_join_fwd / _bwd<E, REPS, GATE, ACC> at every level, and the block forward at dilation 3 over 2 B clips',
                   '# without (MODE 0) and with (MODE 2) the join in its epilogue: k_wrb_conv<C,3,MODE,SAVE,2> / k_nrb_conv<C,3,MODE,SAVE>'] + HEAD + lines) + '\n')
    print(open(os.path.join(ROOT, 'profiles', 'r06_pmc_bwds_C16.txt')).read())
    print(open(os.path.join(ROOT, 'profiles', 'r06_pmc_skip.txt')).read())


if __name__ == '__main__':
    main()
