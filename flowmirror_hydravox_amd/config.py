"""Model hyper-parameters of the HydraVox-CV3 speech-synthesis hot path.

`hydravox.yaml` ships with the (unavailable) weights, so the full-size preset follows
SURVEY.md Appendix A: in-repo evidence where it exists (reference file:line cited there) and
upstream CosyVoice3 values marked [ASSUMED-CV3] otherwise.  Every dimension is a field so tests
can run the same code at toy sizes.
"""
from dataclasses import dataclass, field, asdict
from typing import List


@dataclass
class LLMConfig:
    # Qwen2 backbone (server/model_utils/cosyvoice/llm/llm_multi_head_v3.py:231-260; HF Qwen2)
    hidden: int = 896
    layers: int = 24
    q_heads: int = 14
    kv_heads: int = 2
    head_dim: int = 64
    inter: int = 4864
    rope_theta: float = 1.0e6
    rms_eps: float = 1.0e-6
    text_vocab: int = 151936
    # speech LM (llm_multi_head_v3.py:639-683)
    speech_tokens: int = 6561          # V
    extra_tokens: int = 200            # vocab = V + 200, stop ids = V..V+199
    # MTP heads: Qwen2DecoderLayer(Qwen2Config(hidden_size=H, heads=mtp_heads, kv=mtp_heads), 0)
    head_num: int = 5
    mtp_heads: int = 14
    mtp_inter: int = 22016             # Qwen2Config default (llm_multi_head_v3.py:657-667)
    mtp_rms_eps: float = 1.0e-6

    @property
    def vocab(self) -> int:
        return self.speech_tokens + self.extra_tokens

    @property
    def sos(self) -> int:
        return self.speech_tokens

    @property
    def eos(self) -> int:
        return self.speech_tokens + 1

    @property
    def task_id(self) -> int:
        return self.speech_tokens + 2

    @property
    def mtp_attn_dim(self) -> int:
        # HF: head_dim = hidden // heads ; q/k/v project to heads*head_dim
        return (self.hidden // self.mtp_heads) * self.mtp_heads


@dataclass
class FlowConfig:
    # CausalMaskedDiffWithDiT (cosyvoice/flow/flow.py:278-312)
    vocab: int = 6561
    mel: int = 80
    spk_embed_dim: int = 192
    token_mel_ratio: int = 2
    pre_lookahead_len: int = 3
    pre_lookahead_channels: int = 1024
    # CausalConditionalCFM (cosyvoice/flow/flow_matching.py:197-228)
    n_timesteps: int = 10
    cfg_rate: float = 0.7
    noise_frames: int = 15000
    # DiT estimator (cosyvoice/flow/DiT/dit.py:104-143)
    dim: int = 1024
    depth: int = 22
    heads: int = 16
    head_dim: int = 64
    ff_mult: int = 2
    conv_kernel: int = 31
    conv_groups: int = 16
    time_freq_dim: int = 256
    static_chunk_size: int = 50      # mel frames per chunk of the streaming attention mask (dit.py:119; 2 x token_hop_len 25, cli/model.py:396)

    @property
    def ff(self) -> int:
        return self.dim * self.ff_mult

    @property
    def in_dim(self) -> int:
        return self.mel * 4      # cat[x, cond, mu, spks]


@dataclass
class HiftConfig:
    # CausalHiFTGenerator (cosyvoice/hifigan/generator.py:573-668)
    mel: int = 80
    base_channels: int = 512
    nb_harmonics: int = 8
    sampling_rate: int = 24000
    nsf_alpha: float = 0.1
    nsf_sigma: float = 0.003
    nsf_voiced_threshold: float = 10.0
    upsample_rates: List[int] = field(default_factory=lambda: [8, 5, 3])
    upsample_kernel_sizes: List[int] = field(default_factory=lambda: [16, 11, 7])
    n_fft: int = 16
    hop: int = 4
    resblock_kernel_sizes: List[int] = field(default_factory=lambda: [3, 7, 11])
    resblock_dilations: List[List[int]] = field(default_factory=lambda: [[1, 3, 5], [1, 3, 5], [1, 3, 5]])
    source_resblock_kernel_sizes: List[int] = field(default_factory=lambda: [7, 7, 11])
    source_resblock_dilations: List[List[int]] = field(default_factory=lambda: [[1, 3, 5], [1, 3, 5], [1, 3, 5]])
    lrelu_slope: float = 0.1
    audio_limit: float = 0.99
    conv_pre_look_right: int = 4
    f0_channels: int = 512
    noise_seconds: int = 300           # length of the fixed noise tables (generator.py:226,356)

    @property
    def upsample_total(self) -> int:
        u = self.hop
        for r in self.upsample_rates:
            u *= r
        return u                        # 480 samples per mel frame


@dataclass
class MatchaConfig:
    """Matcha-TTS flow decoder (matcha/models/components/decoder.py:201-300; CosyVoice variant cosyvoice/flow/decoder.py:88-208)."""
    mel: int = 80
    spk_dim: int = 0                    # width of the broadcast speaker vector packed after mu (0: none)
    use_cond: bool = False              # CosyVoice variant: a (B, mel, T) `cond` packed last
    channels: tuple = (256, 256)
    n_blocks: int = 1
    num_mid_blocks: int = 2
    num_heads: int = 2
    head_dim: int = 64
    ff_mult: int = 4                    # FeedForward(mult=4) is fixed in BasicTransformerBlock (transformer.py:236)
    cv_variant: bool = False            # key-padding masks + skip trimming (ConditionalDecoder.forward)
    n_timesteps: int = 10
    temperature: float = 0.667
    max_t: int = 8192

    @property
    def in_channels(self):
        return 2 * self.mel + self.spk_dim + (self.mel if self.use_cond else 0)


@dataclass
class HifiGanConfig:
    """HiFi-GAN v1 (matcha/hifigan/config.py:1-28)."""
    mel: int = 80
    initial_channel: int = 512
    upsample_rates: tuple = (8, 8, 2, 2)
    upsample_kernel_sizes: tuple = (16, 16, 4, 4)
    resblock_kernel_sizes: tuple = (3, 7, 11)
    resblock_dilations: tuple = ((1, 3, 5), (1, 3, 5), (1, 3, 5))
    sampling_rate: int = 22050
    n_fft: int = 1024                   # Denoiser filter_length
    n_overlap: int = 4                  # Denoiser hop = n_fft / n_overlap

    @property
    def upsample(self):
        u = 1
        for r in self.upsample_rates:
            u *= r
        return u


def matcha_config() -> MatchaConfig:
    return MatchaConfig()


def cv2_decoder_config() -> MatchaConfig:
    """CosyVoice-2 ConditionalDecoder dims (cosyvoice/flow/flow.py:36-40): in 320 = x + mu + spks(80) + cond."""
    return MatchaConfig(spk_dim=80, use_cond=True, channels=(256,), n_blocks=4, num_mid_blocks=12, num_heads=8, cv_variant=True)


def tiny_matcha_config(cv=False) -> MatchaConfig:
    if cv:
        return MatchaConfig(spk_dim=80, use_cond=True, channels=(64, 128), n_blocks=1, num_mid_blocks=1, num_heads=2, cv_variant=True,
                            n_timesteps=3, max_t=512)
    return MatchaConfig(spk_dim=64, channels=(64, 64), n_blocks=1, num_mid_blocks=1, num_heads=2, n_timesteps=3, max_t=512)


def tiny_hifigan_config() -> HifiGanConfig:
    return HifiGanConfig(initial_channel=128, upsample_rates=(4, 2), upsample_kernel_sizes=(8, 4), resblock_kernel_sizes=(3, 5),
                         resblock_dilations=((1, 3, 5), (1, 3, 5)), n_fft=64, n_overlap=2)


@dataclass
class HvxConfig:
    llm: LLMConfig = field(default_factory=LLMConfig)
    flow: FlowConfig = field(default_factory=FlowConfig)
    hift: HiftConfig = field(default_factory=HiftConfig)
    sample_rate: int = 24000

    def to_dict(self):
        return asdict(self)


def cv3_config() -> HvxConfig:
    """Full-size HydraVox-CV3."""
    return HvxConfig()


def cv3w_config() -> HvxConfig:
    """HydraVox-CV3 WIDTHS at 2 layers / 2 DiT blocks: every per-layer shape of the full-size model (hidden 896, 14:2 heads, inter 4864,
    vocab 6761, 5 MTP heads of 22016; DiT 1024 x 16 heads, conv groups 16, ff 2048, pre-lookahead 1024; HiFT base 512), so the kernel
    instantiations the benchmark dispatches are the ones the parity tests run, at a depth the CPU reference / oracle finishes in seconds.
    Only the depth and the (gather-only) text vocabulary are reduced."""
    return HvxConfig(
        llm=LLMConfig(layers=2, text_vocab=1024),
        flow=FlowConfig(depth=2),
        hift=HiftConfig(noise_seconds=16),
    )


def cv3d_config() -> HvxConfig:
    """HydraVox-CV3 at FULL DEPTH (24 LM layers, 22 DiT blocks) and full widths; only the gather-only text vocabulary is reduced so that the
    seeded state stays small.  The reference itself runs this on the CPU at short lengths in minutes (tests/golden/make_golden.py: gen_*_cv3d)."""
    return HvxConfig(
        llm=LLMConfig(text_vocab=1024),
        flow=FlowConfig(),
        hift=HiftConfig(noise_seconds=16),
    )


def tiny_config() -> HvxConfig:
    """Toy dimensions used by the parity tests and golden fixtures (same code paths)."""
    return HvxConfig(
        llm=LLMConfig(hidden=128, layers=2, q_heads=2, kv_heads=1, head_dim=64, inter=256,
                      rope_theta=1.0e6, text_vocab=512, speech_tokens=96, extra_tokens=200,
                      head_num=3, mtp_heads=2, mtp_inter=22016),
        # dim/conv_groups must give a multiple of 32 channels per group (MFMA k-step)
        flow=FlowConfig(vocab=96, pre_lookahead_channels=64, dim=512, depth=2, heads=8,
                        head_dim=64, ff_mult=2, conv_groups=16),
        hift=HiftConfig(base_channels=64, f0_channels=64, noise_seconds=8),
    )
