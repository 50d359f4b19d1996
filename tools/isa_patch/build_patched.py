"""Round 4 investigation of the run-to-run different dW2 accumulator of k_nrb_bwd_fused (csrc/conv_wide_bf16.hip, -DTTRAP_NRB_DW2_SLP):
compile conv_wide_bf16.hip with -save-temps, PATCH the device assembly, re-run the rest of hipcc's own pipeline (assembler, lld, offload
bundler, host compile with the patched fat binary embedded) and link variant libraries timbre-trap_amd/lib/libttrap_<tag>.so out of the
shipped objects with the patched conv_wide_bf16.o in place of the original.  Select with TTRAP_LIB=libttrap_<tag>.so.

    python tools/isa_patch/build_patched.py            # builds every variant of PATCHES

Patches act on the k_nrb_bwd_fused<..> kernels only:
  slp0   no patch (the failing form as compiled)
  war    s_nop 1 between a v_pk_fma_f32 and an immediately following v_pk_mov_b32 / v_pk_mul_f32 / v_mov that overwrites one of its sources
  raw    s_nop 1 after every v_pk_mul_f32 (its result feeds packed multiply-adds a few instructions later)
  allpk  s_nop 1 after EVERY v_pk_fma_f32
  mfma   s_nop 7 after every v_mfma in the phase-1 block that holds the dW2 update
"""
import os
import re
import shlex
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, 'timbre-trap_amd', 'csrc')
LIB = os.path.join(ROOT, 'timbre-trap_amd', 'lib')
WORK = '/tmp/ttrap_isa_patch'
HIPCC = '/opt/rocm/bin/hipcc'


def regs(tok):
    m = re.match(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', tok)
    return {int(m.group(1))} if m else set()


def operands(line):
    parts = line.strip().split(None, 1)
    if len(parts) < 2:
        return []
    return [t.strip() for t in parts[1].split(' op_sel')[0].split(',')]


def patch(lines, mode):
    out, inside, n = [], False, 0
    for i, l in enumerate(lines):
        if re.match(r'^_ZN\S*k_nrb_bwd_fused\S*:', l):
            inside = True
        if inside and 's_endpgm' in l:
            inside = False
        out.append(l)
        if not inside or mode == 'slp0':
            continue
        op = l.split()[0] if l.strip() else ''
        nxt = lines[i + 1] if i + 1 < len(lines) else ''
        nop = nxt.split()[0] if nxt.strip() else ''
        if mode == 'allpk' and op == 'v_pk_fma_f32':
            out.append('\ts_nop 1'); n += 1
        elif mode == 'raw' and op == 'v_pk_mul_f32':
            out.append('\ts_nop 1'); n += 1
        elif mode == 'war' and op == 'v_pk_fma_f32' and nop.startswith('v_'):
            srcs = set().union(*[regs(t) for t in operands(l)[1:]])
            dst = regs(operands(nxt)[0]) if operands(nxt) else set()
            if srcs & dst:
                out.append('\ts_nop 1'); n += 1
        elif mode == 'mfma' and op.startswith('v_mfma'):
            out.append('\ts_nop 7'); n += 1
    return out, n


def run(cmd, cwd):
    r = subprocess.run(cmd, cwd=cwd, shell=isinstance(cmd, str), capture_output=True, text=True)
    if r.returncode != 0:
        raise SystemExit('FAILED: %s\n%s' % (cmd if isinstance(cmd, str) else ' '.join(cmd), r.stderr[-3000:]))
    return r


def main():
    os.makedirs(WORK, exist_ok=True)
    for f in os.listdir(WORK):
        os.remove(os.path.join(WORK, f))
    r = subprocess.run([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-DTTRAP_NRB_DW2_SLP', '-I', CSRC, '-c',
                        os.path.join(CSRC, 'conv_wide_bf16.hip'), '-o', 'cw.o', '-save-temps', '-v'], cwd=WORK, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    cmds = [l.strip() for l in r.stderr.split('\n') if l.startswith(' "')]
    dev_s = 'conv_wide_bf16-hip-amdgcn-amd-amdhsa-gfx950.s'
    # the steps after the device assembly was written: device assembler, lld, bundler, host -S (embeds the fat binary), host assembler
    def is_host_S(c):
        t = shlex.split(c)
        return '-cc1' in t and t[t.index('-triple') + 1].startswith('x86_64') and '-S' in t
    tail = [c for c in cmds if ('-cc1as' in c and 'amdgcn' in c) or 'lld"' in c or 'clang-offload-bundler' in c
            or is_host_S(c) or ('-cc1as' in c and 'x86_64' in c)]
    assert len(tail) == 5, [t[:80] for t in tail]
    original = open(os.path.join(WORK, dev_s)).read().split('\n')
    objs = [os.path.join(LIB, f) for f in sorted(os.listdir(LIB)) if f.endswith('.o') and f != 'conv_wide_bf16.o']
    for mode in (sys.argv[1:] or ['slp0', 'war', 'raw', 'allpk', 'mfma']):
        patched, n = patch(original, mode)
        open(os.path.join(WORK, dev_s), 'w').write('\n'.join(patched))
        for c in tail:
            run(c, WORK)
        so = os.path.join(LIB, 'libttrap_%s.so' % mode)
        run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', so, os.path.join(WORK, 'cw.o')] + objs, WORK)
        print('%-6s %4d s_nop inserted -> %s' % (mode, n, so))


if __name__ == '__main__':
    main()
