// stft_victim.hip — the product's hift_stft_kernel (csrc/hift_ops.hip, included verbatim) behind a C entry point, for tools/mfma_interference.py
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -shared -fPIC -I flowmirror_hydravox_amd/csrc -I include -o tools/bin/libstft.so tools/stft_victim.hip
#include <stdarg.h>
#include "../flowmirror_hydravox_amd/csrc/hift_ops.hip"
namespace hvx { void set_error(const char*, ...) {} }
extern "C" int stft_launch(const float* x, float* spec, int L, void* stream) { return hvx::launch_hift_stft(x, spec, L, 32, (hipStream_t)stream); }
extern "C" int istft_launch(const float* x, float* wav, int frames, void* stream) { return hvx::launch_hift_istft(x, 32, wav, frames, 0.99f, (hipStream_t)stream); }
