"""Group a tools/prof_summary.py listing into kernel families: python tools/prof_families.py summary.txt"""
import collections
import re
import sys

GROUPS = [('bf16 block forward (k_wrb_conv / k_nrb_conv MODE 0)', r'k_[wn]rb_conv.*ELi0ELb'), ('bf16 block data gradient (MODE 1)', r'k_[wn]rb_conv.*ELi1ELb'),
          ('bf16 block data + weight gradient in one pass (k_wrb_dxw / k_nrb_dxw)', r'k_[wn]rb_dxw'), ('bf16 one-pass strip backward, h1 saved (k_wrb_bwds + weight prep)', r'k_wrb_bwds|k_lvl_wprep'), ('bf16 one-pass backward with recomputed h1 (opt-in)', r'k_wrb_bwd_fused'),
          ('bf16 block pointwise backward (bwd_a)', r'k_[wn]rb_bwd_a'), ('bf16 block weight gradient', r'k_[wn]rb_wgrad'),
          ('bf16 narrow fused backward', r'k_nrb_bwd_fused'), ('bf16 partial-sum reduces', r'k_[wn]rb_reduce|k_w4_reduce|k_lat_wred|k_edge_reduce'),
          ('bf16 strided / transposed layers (k_s4 / k_p2)', r'k_s4|k_p2'), ('bf16 strided weight gradient (k_w4)', r'k_w4<'),
          ('bf16 latent heads', r'k_lat_'), ('bf16 boundary convs', r'k_cin_|k_cout_'), ('fp32 <-> bf16 layout changes', r'k_wide_(un)?pack'),
          ('narrow fused backward (k_small_bwd_fused)', r'k_small_bwd_fused|k_small_wgrad_reduce'), ('narrow resblocks fwd/dgrad (k_small)', r'k_small<|k_small_lds|k_small_fwd4'), ('narrow pointwise backward (k_small_bwd_a)', r'k_small_bwd_a'),
          ('weight grad 3x3 C>=16', r'k_wgrad_dma<(16|32), (16|32), 16, .*WRes'), ('weight grad other', r'k_wgrad3_pack_reduce'), ('weight grad 3x3 C<=8', r'k_wgrad_dma<(4|8), (4|8), .*WRes|k_wgrad3_pack<'),
          ('weight grad strided', r'k_wgrad_dma<.*WStr'), ('weight grad other', r'k_wgrad_reduce|k_wgrad_generic|k_wgrad3x3|k_wgrad_mfma|k_wgrad3_pack_reduce'),
          ('wide resblocks forward (k_rb_fwd)', r'k_rb_fwd'), ('wide 3x3 data gradient', r'k_conv_mfma<(16|32), (16|32), .*Res3x3'),
          ('wide pointwise backward (k_rb_bwd_a)', r'k_rb_bwd_a'), ('strided / transposed convs', r'k_conv_mfma<.*(Up4|Down4)|k_conv_valu'),
          ('latent GEMMs', r'k_gemm'), ('gate / bias passes', r'k_gate|k_elu_bwd|k_channel_sum|k_gated'),
          ('boundary convs', r'k_conv3x3_small|k_conv3x3_lds|k_conv_generic'), ('torch elementwise', r'at::native'),
          ('CQT', r'k_band|k_fft|k_spec|k_scale'), ('losses / optimizer', r'k_sqdiff|k_trn|k_act|k_adamw|k_l2|k_scaled|k_dot|k_sumsq|k_finalize|k_window_ola')]
acc = collections.OrderedDict((g, 0.0) for g, _ in GROUPS)
other = 0.0
for line in open(sys.argv[1]):
    m = re.match(r'\s*([\d.]+) ms/step.*avg=\s*[\d.]+ us\s+(.*)', line)
    if not m:
        continue
    for g, pat in GROUPS:
        if re.search(pat, m.group(2)):
            acc[g] += float(m.group(1))
            break
    else:
        other += float(m.group(1))
for g, v in acc.items():
    print('%-44s %6.1f ms/step' % (g, v))
print('%-44s %6.1f ms/step' % ('other', other))
print('%-44s %6.1f ms/step' % ('total', sum(acc.values()) + other))
