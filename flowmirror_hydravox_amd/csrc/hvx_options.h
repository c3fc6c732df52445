// hvx_options.h — the library's run-time options: ONE mechanism, visible to the C-ABI caller (hvx_set_option / hvx_get_option / hvx_option_name in
// include/hvx.h), instead of environment variables read through function-local statics.  None of them changes a result's CONTRACT: they pick between forms
// the parity tests hold to each other (dec_gemm, conv64_resident, x3p8, ...) or tune a launch geometry.  Numerics that a caller may care about are NOT here:
// the vocoder's exact-fp32 form is a field of hvx_hift_config / hvx_hifigan_config.  Options marked `lab` exist for timing studies (tools/): the shipped
// library refuses to set them — they are settable only in a library compiled with -DHVX_LAB (python -m flowmirror_hydravox_amd.build --lab ...).
// Values are read at dispatch time (an atomic load per launch decision); a captured step graph keeps the forms it was captured with.
#pragma once

#if (defined(HVX_LAB_GEMM_EPI) || defined(HVX_LAB_X3_EPI) || defined(HVX_ATTN_LAB)) && !defined(HVX_LAB)
#error "timing-only kernel variants (results are not stored) need -DHVX_LAB: the loader then refuses the library unless HVX_LIB_PATH names it"
#endif

namespace hvx {

enum OptId : int {
    OPT_ATT_CHUNK = 0,         // keys per split of the decode attention; 0 = chosen per grid (hvx_llm.hip); else a multiple of 128
    OPT_ATT_WAVES,             // 8: the one-tile bf16 decode attention merges 8 waves per split workgroup; 4: four
    OPT_GEMM_BIG_GW,           // column tiles per row group of the 256-tile Linear's workgroup order; 0 = all columns in one group
    OPT_GEMM_BIG_MIN_TILES,    // fewest 256 x 256 tiles for which the 256-tile Linear is chosen over the 128-tile one
    OPT_DEC_GEMM,              // 1: 33..256-row decode grids on gemm_dec.hip (fragment-order activations); 0: the generic skinny kernels (the tests compare the two)
    OPT_DEC_HEADS,             // bit 0: MTP head MLPs, bit 1: the shared output projection on gemm_dec.hip
    OPT_CONV_RESIDENT,         // 1: the DiT position embedding's 64-channel grouped convolution in the resident-row form; 0: tiled
    OPT_CONV64_RESIDENT,       // 1: the vocoder's 64-channel split-bf16 convolutions in the resident-row form; 0: tiled (the tests compare the two)
    OPT_X3P8,                  // 1: the 8-wave 128 x 128 x 64 tile for the vocoder's 128- / 256-channel split-bf16 convolutions; 0: the 4-wave tile
    OPT_ATTN_DIT_FORM,         // DiT attention tile: 0 = chosen per shape (16x16x32 MFMAs; the pipeline rotated across key tiles where that form exists), 17 = the same,
                               // 16 = the in-tile pipeline everywhere (the round-5 tile), 32 = the 32x32x16 tile (measured slower, kept selectable)
    OPT_DEC_FUSE_ROWS,         // decode grids of up to this many rows build the o_proj's activation fragments from the attention's key-split partials (no combine launch);
                               // 0 = never (the default: bit-identical and measured 12 % SLOWER per step, profiles/r06_decode_fuse_ab.log — kept selectable for that A / B)
    // ---- lab (settable with -DHVX_LAB only) ----
    OPT_GEMM_BIG_MFMA,         // MFMA shape of the 256-tile Linear's K-loop: 16 (16x16x32) or 32 (32x32x16: bit-identical, measured 4 % slower)
    OPT_HEAD_DOWN_SPLIT,       // forced K split of the MTP heads' down projection; 0 = chosen per grid
    OPT_DEC_GPW_QKV, OPT_DEC_GPW_RES, OPT_DEC_GPW_MLP, OPT_DEC_GPW_DOWN, OPT_DEC_GPW_OUT, OPT_DEC_GPW_HMLP,   // column groups per workgroup of the decode GEMM launches
    OPT_ATTN_LAB,              // timing-only variants of the DiT attention loop (results are garbage): bit mask, see attention.hip
    OPT_ATTN_NW,               // 8: one 8-wave workgroup per CU for the DiT attention (lab)
    OPT_COUNT
};

struct OptDef { const char* name; long long dflt; int lab; };
extern const OptDef g_opt_defs[OPT_COUNT];
long long opt(OptId id);       // current value (relaxed atomic load)

}  // namespace hvx
