"""ctypes binding of libhvx.so (include/hvx.h).  The product path fails loudly when the library or a GPU
is missing: there is no CPU fallback anywhere in this package."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# (HVX_LIB_PATH: lab builds only — A / B of compiler flags on the same box, `python -m flowmirror_hydravox_amd.build --lab NAME` prints the path to put here; the
# product always loads the in-tree libhvx.so, and load() refuses a library built with -DHVX_LAB unless it was named this way)
LIB_PATH = os.environ.get('HVX_LIB_PATH') or os.path.join(HERE, 'libhvx.so')
LIB_EXPLICIT = bool(os.environ.get('HVX_LIB_PATH'))

F32, BF16 = 0, 1

# Act codes (csrc/hvx_device.h)
ACT_NONE, ACT_GELU_TANH, ACT_SILU, ACT_MISH, ACT_ELU, ACT_LRELU, ACT_SNAKE, ACT_TANH, ACT_ABS = range(9)

c_i32, c_i64, c_f32, c_vp, c_sz = C.c_int32, C.c_int64, C.c_float, C.c_void_p, C.c_size_t


class HvxError(RuntimeError):
    pass


class SampleArgs(C.Structure):
    _fields_ = [('n_seq', c_i32), ('head_k', c_i32), ('vocab', c_i32), ('speech_tokens', c_i32),
                ('logp', c_vp), ('logp_seq_stride', c_i64), ('logp_head_stride', c_i64),
                ('hist', c_vp), ('hist_seq_stride', c_i64), ('hist_len', c_vp),
                ('min_len', c_vp), ('active', c_vp),
                ('top_k', c_i32), ('top_p', c_f32), ('win_size', c_i32), ('rep_thresh', c_i32),
                ('noise', c_vp), ('noise_seq_stride', c_i64), ('noise_len', c_i32),
                ('cursor', c_vp), ('out_ids', c_vp), ('max_trials', c_i32), ('noise_limit', c_vp)]


class GemmArgs(C.Structure):
    _fields_ = [('dtype', c_i32), ('M', c_i32), ('N', c_i32), ('K', c_i32), ('batch', c_i32), ('groups', c_i32),
                ('A', c_vp), ('a_bs', c_i64), ('lda', c_i32), ('a_gs', c_i32), ('rows_in', c_i32),
                ('cin_pad', c_i32), ('conv_stride', c_i32), ('conv_dil', c_i32), ('pad_left', c_i32), ('up', c_i32),
                ('W', c_vp), ('w_gs', c_i64),
                ('bias', c_vp),
                ('act', c_i32), ('act_param', c_f32), ('act_alpha', c_vp),
                ('gate', c_vp), ('gate_bs', c_i64),
                ('res', c_vp), ('res_bs', c_i64), ('ldres', c_i32), ('res_row_off', c_i32),
                ('scale', c_f32),
                ('out', c_vp), ('out_f32', c_i32), ('out_bs', c_i64), ('ldo', c_i32), ('out_row_off', c_i32), ('out_cols', c_i32),
                ('out2', c_vp), ('act2', c_i32), ('act2_param', c_f32), ('act2_alpha', c_vp), ('out2_bs', c_i64),
                ('ldo2', c_i32), ('out2_row_off', c_i32), ('out2_cols', c_i32), ('x3', c_i32), ('res_f16', c_i32), ('out_f16', c_i32)]


class AttnArgs(C.Structure):
    _fields_ = [('dtype', c_i32), ('batch', c_i32), ('heads', c_i32), ('t', c_i32), ('t_pad', c_i32),
                ('q', c_vp), ('k', c_vp), ('vT', c_vp), ('out', c_vp), ('kv_len', c_vp),
                ('causal', c_i32), ('scale', c_f32),
                ('n_splits', c_i32), ('split_chunk', c_i32), ('part_o', c_vp), ('part_ml', c_vp), ('chunk', c_i32), ('q_log2', c_i32)]


class FeatureConfig(C.Structure):
    _fields_ = [('frame_len', c_i32), ('hop', c_i32), ('reflect_pad', c_i32), ('n_frames', c_i32), ('bins', c_i32), ('power', c_i32), ('mag_eps', c_f32),
                ('n_mels', c_i32), ('log_floor', c_f32), ('log_scale', c_f32), ('post', c_i32), ('time_major', c_i32)]


class LLMConfig(C.Structure):
    _fields_ = [('dtype', c_i32), ('hidden', c_i32), ('layers', c_i32), ('q_heads', c_i32), ('kv_heads', c_i32), ('inter', c_i32),
                ('vocab', c_i32), ('vocab_pad', c_i32), ('speech_tokens', c_i32), ('text_vocab', c_i32),
                ('head_num', c_i32), ('mtp_attn_dim', c_i32), ('mtp_inter', c_i32),
                ('rms_eps', c_f32), ('mtp_rms_eps', c_f32), ('max_pos', c_i32)]


class DecodeArgs(C.Structure):
    _fields_ = [('n_seq', c_i32), ('head_k', c_i32), ('win_cap', c_i32), ('max_out', c_i32),
                ('tok', c_vp), ('ctrl', c_vp), ('hist', c_vp), ('hist_len', c_vp), ('min_adj', c_vp), ('active', c_vp),
                ('seq_state', c_vp), ('out_tokens', c_vp), ('ids', c_vp), ('logp', c_vp),
                ('top_k', c_i32), ('top_p', c_f32), ('win_size', c_i32), ('rep_thresh', c_i32), ('max_trials', c_i32),
                ('noise', c_vp), ('noise_seq_stride', c_i64), ('noise_len', c_i32), ('cursor', c_vp), ('noise_limit', c_vp)]


class FlowConfig(C.Structure):
    _fields_ = [('dtype', c_i32), ('vocab', c_i32), ('mel', c_i32), ('spk_dim', c_i32), ('pla_channels', c_i32), ('pla_len', c_i32),
                ('dim', c_i32), ('depth', c_i32), ('heads', c_i32), ('ff', c_i32), ('conv_kernel', c_i32), ('conv_groups', c_i32),
                ('time_freq_dim', c_i32), ('max_t', c_i32), ('cfg_rate', c_f32)]


class HiftConfig(C.Structure):
    _fields_ = [('mel', c_i32), ('base_channels', c_i32), ('nb_harmonics', c_i32), ('f0_channels', c_i32),
                ('n_up', c_i32), ('up_rates', c_i32 * 4), ('up_kernels', c_i32 * 4),
                ('n_rb', c_i32), ('rb_kernels', c_i32 * 4), ('rb_dils', (c_i32 * 3) * 4),
                ('src_rb_kernels', c_i32 * 4), ('src_rb_dils', (c_i32 * 3) * 4),
                ('n_fft', c_i32), ('hop', c_i32), ('conv_pre_kernel', c_i32), ('conv_post_kernel', c_i32),
                ('sampling_rate', c_f32), ('nsf_alpha', c_f32), ('nsf_sigma', c_f32), ('voiced_threshold', c_f32),
                ('lrelu_slope', c_f32), ('audio_limit', c_f32), ('exact_fp32', c_i32)]


class MatchaConfig(C.Structure):
    _fields_ = [('in_channels', c_i32), ('out_channels', c_i32), ('n_stages', c_i32), ('channels', c_i32 * 4),
                ('n_blocks', c_i32), ('n_mid', c_i32), ('heads', c_i32), ('ff_mult', c_i32), ('cv_variant', c_i32), ('max_t', c_i32)]


class HifiGanConfig(C.Structure):
    _fields_ = [('mel', c_i32), ('initial_channel', c_i32), ('n_up', c_i32), ('up_rates', c_i32 * 4), ('up_kernels', c_i32 * 4),
                ('n_rb', c_i32), ('rb_kernels', c_i32 * 4), ('rb_dils', (c_i32 * 3) * 4), ('exact_fp32', c_i32)]


class NdDesc(C.Structure):
    _fields_ = [('ndim', c_i32), ('shape', c_i32 * 6), ('stride_a', c_i64 * 6), ('stride_b', c_i64 * 6), ('stride_c', c_i64 * 6)]


# every symbol include/hvx.h declares: name -> (restype, argtypes)
SYMBOLS = {
    'hvx_abi_version': (c_i32, []),
    'hvx_last_error': (C.c_char_p, []),
    'hvx_device_ok': (c_i32, []),
    'hvx_set_option': (c_i32, [C.c_char_p, c_i64]),
    'hvx_get_option': (c_i32, [C.c_char_p, C.POINTER(c_i64)]),
    'hvx_option_name': (C.c_char_p, [c_i32, C.POINTER(c_i32), C.POINTER(c_i64)]),
    'hvx_build_flags': (C.c_char_p, []),
    'hvx_is_lab_build': (c_i32, []),
    'hvx_stream_create_cu_range': (c_i32, [c_i32, c_i32, C.POINTER(c_vp)]),
    'hvx_stream_destroy': (c_i32, [c_vp]),
    'hvx_prof_enable': (c_i32, [c_i32]),
    'hvx_prof_read': (c_i32, [c_i32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(c_i64), C.POINTER(c_i64), C.POINTER(C.c_double)]),
    'hvx_ras_sample': (c_i32, [C.POINTER(SampleArgs), c_vp]),
    'hvx_op_gemm': (c_i32, [C.POINTER(GemmArgs), c_vp]),
    'hvx_op_attention': (c_i32, [C.POINTER(AttnArgs), c_vp]),
    'hvx_op_resample_linear': (c_i32, [c_vp, c_i32, c_i32, c_vp, c_i32, c_vp]),
    'hvx_op_skinny_gemm': (c_i32, [c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_vp, c_vp, c_i32, c_vp, c_vp, c_i32, c_vp]),
    'hvx_llm_create': (c_i32, [C.POINTER(LLMConfig), C.POINTER(c_vp), c_i32, C.POINTER(c_vp)]),
    'hvx_llm_destroy': (None, [c_vp]),
    'hvx_llm_workspace_bytes': (c_sz, [c_vp, c_i32, c_i32, c_i32]),
    'hvx_llm_kv_bytes': (c_sz, [c_vp, c_i32, c_i32]),
    'hvx_llm_bind': (c_i32, [c_vp, c_vp, c_sz, c_i32, c_i32, c_vp, c_sz, c_i32, c_i32, c_vp]),
    'hvx_llm_forward': (c_i32, [c_vp, c_vp, c_i32, c_i32, c_vp, c_vp, c_i32, c_vp]),
    'hvx_llm_decode_steps': (c_i32, [c_vp, c_vp, C.POINTER(DecodeArgs), c_i32]),
    'hvx_llm_decode_join': (c_i32, [c_vp, c_vp, C.POINTER(DecodeArgs), c_i32, c_i32, c_i32, c_i32, c_i32]),
    'hvx_llm_use_graph': (c_i32, [c_vp, c_i32]),
    'hvx_llm_set_head_mlp_fp8': (c_i32, [c_vp, c_vp, c_vp]),
    'hvx_llm_last_hidden': (c_i32, [c_vp, c_vp, c_i32, c_vp]),
    'hvx_flow_create': (c_i32, [C.POINTER(FlowConfig), C.POINTER(c_vp), c_i32, C.POINTER(c_vp)]),
    'hvx_flow_destroy': (None, [c_vp]),
    'hvx_flow_workspace_bytes': (c_sz, [c_vp, c_i32, c_i32]),
    'hvx_flow_encode': (c_i32, [c_vp, c_vp, c_vp, c_sz, c_vp, c_i32, c_vp, c_vp, c_vp]),
    'hvx_flow_prelookahead': (c_i32, [c_vp, c_vp, c_vp, c_sz, c_vp, c_i32, c_vp]),
    'hvx_cfm_estimator': (c_i32, [c_vp, c_vp, c_vp, c_sz, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'hvx_cfm_estimator_streaming': (c_i32, [c_vp, c_vp, c_vp, c_sz, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp]),
    'hvx_flow_prelookahead_context': (c_i32, [c_vp, c_vp, c_vp, c_sz, c_vp, c_i32, c_vp]),
    'hvx_flow_encode_chunk': (c_i32, [c_vp, c_vp, c_vp, c_sz, c_vp, c_i32, c_vp, c_i32, c_vp, c_vp]),
    'hvx_cfm_solve_streaming': (c_i32, [c_vp, c_vp, c_vp, c_sz, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, C.POINTER(c_f32), C.POINTER(c_f32), c_i32]),
    'hvx_flow_set_mod_cache': (c_i32, [c_vp, c_vp, c_sz]),
    'hvx_flow_set_half_stream': (c_i32, [c_vp, c_i32]),
    'hvx_flow_set_f16_linears': (c_i32, [c_vp, c_i32]),
    'hvx_flow_set_f32_small': (c_i32, [c_vp, c_i32]),
    'hvx_cfm_solve_batch': (c_i32, [c_vp, c_vp, c_vp, c_sz, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, C.POINTER(c_f32), C.POINTER(c_f32), c_i32]),
    'hvx_cfm_solve': (c_i32, [c_vp, c_vp, c_vp, c_sz, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32, C.POINTER(c_f32), C.POINTER(c_f32)]),
    'hvx_matcha_create': (c_i32, [C.POINTER(MatchaConfig), C.POINTER(c_vp), c_i32, C.POINTER(c_vp)]),
    'hvx_matcha_destroy': (None, [c_vp]),
    'hvx_matcha_workspace_bytes': (c_sz, [c_vp, c_i32, c_i32]),
    'hvx_matcha_estimator': (c_i32, [c_vp, c_vp, c_vp, c_sz, c_i32, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp]),
    'hvx_matcha_solve': (c_i32, [c_vp, c_vp, c_vp, c_sz, c_i32, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_i32, C.POINTER(c_f32), C.POINTER(c_f32)]),
    'hvx_hifigan_create': (c_i32, [C.POINTER(HifiGanConfig), C.POINTER(c_vp), c_i32, C.POINTER(c_vp)]),
    'hvx_hifigan_destroy': (None, [c_vp]),
    'hvx_hifigan_workspace_bytes': (c_sz, [c_vp, c_i32]),
    'hvx_hifigan_forward': (c_i32, [c_vp, c_vp, c_vp, c_sz, c_vp, c_i32, c_vp]),
    'hvx_denoise_workspace_bytes': (c_sz, [c_i32, c_i32, c_i32]),
    'hvx_frame_features_workspace_bytes': (c_sz, [c_i32, c_vp]),
    'hvx_frame_features': (c_i32, [c_vp, c_vp, c_sz, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp]),
    'hvx_mel_workspace_bytes': (c_sz, [c_i32, c_i32, c_i32, c_i32]),
    'hvx_mel_spectrogram': (c_i32, [c_vp, c_vp, c_sz, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_i32, c_vp]),
    'hvx_stft_magnitude': (c_i32, [c_vp, c_vp, c_sz, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp]),
    'hvx_denoise': (c_i32, [c_vp, c_vp, c_sz, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_f32, c_vp]),
    'hvx_hift_create': (c_i32, [C.POINTER(HiftConfig), C.POINTER(c_vp), c_i32, C.POINTER(c_vp)]),
    'hvx_hift_destroy': (None, [c_vp]),
    'hvx_hift_set_weight_planes': (c_i32, [c_vp, c_vp, c_i32]),
    'hvx_hift_workspace_bytes': (c_sz, [c_vp, c_i32]),
    'hvx_hift_f0': (c_i32, [c_vp, c_vp, c_vp, c_sz, c_vp, c_i32, c_vp]),
    'hvx_hift_source': (c_i32, [c_vp, c_vp, c_vp, c_sz, c_vp, c_i32, c_vp, c_vp]),
    'hvx_hift_decode': (c_i32, [c_vp, c_vp, c_vp, c_sz, c_vp, c_vp, c_i32, c_vp]),
    'hvx_hift_decode_chunk': (c_i32, [c_vp, c_vp, c_vp, c_sz, c_vp, c_vp, c_i32, c_i32, c_vp]),
    'hvx_nd_elementwise': (c_i32, [c_i32, C.POINTER(NdDesc), c_vp, c_vp, c_vp, C.c_float, C.c_float, c_vp, c_vp]),
    'hvx_rows_reduce': (c_i32, [c_i32, c_vp, c_i64, c_i64, c_vp, c_vp]),
    'hvx_rows_softmax': (c_i32, [c_vp, c_i64, c_i64, c_vp, c_vp]),
    'hvx_avgpool_rows': (c_i32, [c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_vp]),
    'hvx_conv2d': (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
}

# include/hvx.h: HVX_ABI_VERSION — bumped whenever a symbol is added or an argument struct changes size / meaning (2: round 4's six symbols, larger
# SkinnyArgs / AttnArgs, fragment-order KV cache; 3: round 5; 4: round 6 — hvx_set_option / hvx_get_option / hvx_option_name / hvx_build_flags / hvx_is_lab_build,
# `exact_fp32` in hvx_hift_config and hvx_hifigan_config)
HVX_ABI_VERSION = 4

_lib = None


def load():
    """Load libhvx.so and bind every declared symbol.  Raises if the library was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HvxError('libhvx.so is missing at %s — run `python -m flowmirror_hydravox_amd.build` '
                       '(hipcc --offload-arch=gfx950); there is no CPU fallback' % LIB_PATH)
    # PyTorch-ROCm ships its own libamdhip64 (same SONAME as /opt/rocm's).  Import torch first so that libhvx binds to the
    # HIP runtime instance torch already initialised: streams and device pointers are then shared, and there is exactly one
    # runtime in the process (loaded the other way round, the system runtime comes up first and sees no device).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    rebuild = 'rebuild it: `python -m flowmirror_hydravox_amd.build --force`'
    # the version first: a stale library fails HERE with a message, not later on a missing symbol or a struct of another size
    try:
        lib.hvx_abi_version.restype = c_i32
        got = int(lib.hvx_abi_version())
    except AttributeError:
        raise HvxError('%s is not a libhvx (no hvx_abi_version); %s' % (LIB_PATH, rebuild))
    if got != HVX_ABI_VERSION:
        raise HvxError('libhvx ABI version mismatch: %s reports %d, this package binds %d (include/hvx.h); %s' % (LIB_PATH, got, HVX_ABI_VERSION, rebuild))
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise HvxError('libhvx.so does not export %s although its ABI version is %d; %s' % (name, got, rebuild))
        fn.restype = res
        fn.argtypes = args
    if int(lib.hvx_is_lab_build()) and not LIB_EXPLICIT:
        # a library compiled with -DHVX_LAB can reach timing-only kernels that store no results: never by accident
        raise HvxError('%s was built with -DHVX_LAB (flags: %s): a lab library is loaded only when HVX_LIB_PATH names it; %s'
                       % (LIB_PATH, (lib.hvx_build_flags() or b'').decode(), rebuild))
    _lib = lib
    return lib


def set_option(key, value):
    """hvx_set_option (csrc/hvx_options.h): the library's one switchboard — there are no environment variables"""
    check(load().hvx_set_option(key.encode(), int(value)), 'hvx_set_option(%s)' % key)


def get_option(key):
    v = c_i64()
    check(load().hvx_get_option(key.encode(), C.byref(v)), 'hvx_get_option(%s)' % key)
    return int(v.value)


def options():
    """{name: (value, default, lab_only)} of every run-time option of the loaded library"""
    lib, out, i = load(), {}, 0
    while True:
        lab, dflt = c_i32(), c_i64()
        name = lib.hvx_option_name(i, C.byref(lab), C.byref(dflt))
        if not name:
            return out
        out[name.decode()] = (get_option(name.decode()), int(dflt.value), bool(lab.value))
        i += 1


class option_scope:
    """with option_scope(dec_gemm=0): ...  — sets options for the block and restores the previous values (A / B inside one process)"""

    def __init__(self, **kv):
        self.kv, self.old = kv, {}

    def __enter__(self):
        for k, v in self.kv.items():
            self.old[k] = get_option(k)
            set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            set_option(k, v)
        return False


def check(rc, what=''):
    if rc != 0:
        msg = load().hvx_last_error()
        raise HvxError('%s failed: %s' % (what or 'hvx call', msg.decode() if msg else 'unknown error'))


def require_gpu():
    lib = load()
    import torch
    if not torch.cuda.is_available():
        raise HvxError('no ROCm device visible: the HydraVox hot path runs on MI355X only (no CPU fallback)')
    torch.cuda.init()
    torch.zeros(1, device='cuda')          # make sure the HIP primary context exists before libhvx queries it
    n = lib.hvx_device_ok()
    if n <= 0:
        msg = lib.hvx_last_error()
        raise HvxError('libhvx cannot use the current device: %s' % (msg.decode() if msg else 'no gfx950 device'))
    return n


def ptr(t):
    """Device/host pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def cu_range_stream(first_cu, n_cus, device=None):
    """a torch stream over a HIP stream confined to CUs [first_cu, first_cu + n_cus) (hvx_stream_create_cu_range); lives as long as the process"""
    import torch
    h = c_vp()
    check(load().hvx_stream_create_cu_range(int(first_cu), int(n_cus), C.byref(h)), 'hvx_stream_create_cu_range')
    return torch.cuda.ExternalStream(h.value, device=device)


def dtype_code(torch_dtype):
    import torch
    if torch_dtype == torch.float32:
        return F32
    if torch_dtype == torch.bfloat16:
        return BF16
    raise HvxError('unsupported dtype %s (hvx computes in bf16 or f32)' % torch_dtype)


def ptr_array(tensors):
    arr = (c_vp * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr
