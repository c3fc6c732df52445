"""flowmirror_hydravox_amd — MI355X-native HydraVox speech-synthesis hot path.

Host side (this package) mirrors the reference's Python inference surface
(`llm.inference -> flow.inference -> hift.inference`, `ModelManager.load_models/load_pt`,
`worker_process_tts`); all hot-path math runs in hand-written HIP kernels (gfx950) behind the
C-ABI shared library declared in `include/hvx.h`.
"""
from .config import HvxConfig, LLMConfig, FlowConfig, HiftConfig, cv3_config, tiny_config  # noqa: F401

__version__ = "0.1.0"
