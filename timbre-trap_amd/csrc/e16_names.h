// Names of the C entry points of the 16-bit channel-innermost kernels in the fp16 build (-DTT_F16, see bf16_common.h): every
// function the five sources define gets the suffix _h, so that both builds link into one library.  Included before
// include/ttrap.h, whose declarations of these names then declare the _h functions in this translation unit.
#pragma once
#if defined(TT_F16)
#define tt_wide_scratch_bytes tt_wide_scratch_bytes_h
#define tt_wide_pack tt_wide_pack_h
#define tt_wide_unpack tt_wide_unpack_h
#define tt_wide_rb_fwd tt_wide_rb_fwd_h
#define tt_wide_rb_fwd_join tt_wide_rb_fwd_join_h
#define tt_wide_rb_bwd_is_onepass tt_wide_rb_bwd_is_onepass_h
#define tt_wide_rb_bwd tt_wide_rb_bwd_h
#define tt_wide_fused_scratch_bytes tt_wide_fused_scratch_bytes_h
#define tt_wide_rb_bwd_fused tt_wide_rb_bwd_fused_h
#define tt_wide_onepass_scratch_bytes tt_wide_onepass_scratch_bytes_h
#define tt_wide_rb_bwd_onepass tt_wide_rb_bwd_onepass_h
#define tt_stride16_scratch_bytes tt_stride16_scratch_bytes_h
#define tt_sconv16_fwd tt_sconv16_fwd_h
#define tt_sconv16_bwd tt_sconv16_bwd_h
#define tt_tconv16_fwd tt_tconv16_fwd_h
#define tt_tconv16_bwd tt_tconv16_bwd_h
#define tt_latent16_scratch_bytes tt_latent16_scratch_bytes_h
#define tt_latent16_contract tt_latent16_contract_h
#define tt_latent16_expand tt_latent16_expand_h
#define tt_latent16_wgrad tt_latent16_wgrad_h
#define tt_edge16_scratch_bytes tt_edge16_scratch_bytes_h
#define tt_convin16_fwd tt_convin16_fwd_h
#define tt_convin16_bwd tt_convin16_bwd_h
#define tt_convout16_fwd tt_convout16_fwd_h
#define tt_convout16_bwd tt_convout16_bwd_h
#define ttx_red_defer ttx_red_defer_h
#define ttx_wprep_done ttx_wprep_done_h
#define ttx_gate_dx ttx_gate_dx_h
#define ttx_skip ttx_skip_h
#define tt_wide_level_bwd_gated_join tt_wide_level_bwd_gated_join_h
#define ttx_wide_wprep_batch ttx_wide_wprep_batch_h
#define tt_wide_level_bwd tt_wide_level_bwd_h
#define tt_wide_level_bwd_gated tt_wide_level_bwd_gated_h
#define tt_gate16 tt_gate16_h
#define tt_latent16_expand_gated tt_latent16_expand_gated_h
#define tt_latent16_contract_pregated tt_latent16_contract_pregated_h
#define tt_latent16_wgrad_pregated tt_latent16_wgrad_pregated_h
#define tt_latent16_pregated_ok tt_latent16_pregated_ok_h
#define tt_sconv16_bwd_pregated tt_sconv16_bwd_pregated_h
#define tt_tconv16_bwd_pregated tt_tconv16_bwd_pregated_h
#define tt_wide_level_scratch_bytes tt_wide_level_scratch_bytes_h
#endif
