#!/usr/bin/env python3
"""Does a stream of bf16 MFMAs on one HIP stream change the results of an unrelated kernel on another?  No libhvx involved: the aggressor
is agg_mfma (tools/aggressors.hip: MFMAs on registers, no LDS, one load and one store per thread), the victims are vic_kernel's
single-instruction-class loops.   python tools/mfma_interference.py [--aggressor 4] [--launches 20000]"""
import argparse
import ctypes as C
import os
import threading

import torch

ap = argparse.ArgumentParser()
ap.add_argument('--aggressor', type=int, default=4)
ap.add_argument('--agg-blocks', type=int, default=66)
ap.add_argument('--agg-iters', type=int, default=200)
ap.add_argument('--launches', type=int, default=20000)
ap.add_argument('--blocks', type=int, default=66)
ap.add_argument('--iters', type=int, default=64)
ap.add_argument('--stft-lib', default='libstft.so')
ap.add_argument('--only', type=int, default=-1)
a = ap.parse_args()
dev = torch.device('cuda', 0)
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'bin', 'libagg.so'))
lib.agg_launch.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
lib.vic_launch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
stop = threading.Event()
go = threading.Event()


def aggressor():
    torch.cuda.set_device(dev)
    s = torch.cuda.Stream(device=dev)
    buf = torch.rand(a.agg_blocks * 256, device=dev)
    torch.cuda.synchronize()
    go.set()
    while not stop.is_set():
        for _ in range(50):
            assert lib.agg_launch(a.aggressor, buf.data_ptr(), a.agg_blocks, a.agg_iters, s.cuda_stream) == 0
        s.synchronize()


stft = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'bin', a.stft_lib))
stft.stft_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
names = ['fp32 FMA chain', 'packed fp32 FMA chain', 'fp32 MFMA chain', 'bf16 MFMA chain', 'LDS table reads + FMA', 'exp2 / sin', 'hift_stft_kernel (product)']
sv = torch.cuda.Stream(device=dev)
x = torch.rand(a.blocks * 128, device=dev)
for with_agg in (False, True):
    if with_agg:
        th = threading.Thread(target=aggressor)
        th.start()
        go.wait()
    for kind in (range(7) if a.only < 0 else [a.only]):
        out = torch.empty_like(x) if kind < 6 else torch.empty(33600 // 4 + 1, 32, device=dev)
        src = torch.tanh(0.3 + 0.01 * torch.randn(33600, device=dev))
        with torch.cuda.stream(sv):
            def launch():
                if kind < 6:
                    lib.vic_launch(kind, x.data_ptr(), out.data_ptr(), a.blocks, a.iters, sv.cuda_stream)
                else:
                    stft.stft_launch(src.data_ptr(), out.data_ptr(), 33600, sv.cuda_stream)
            launch()
            sv.synchronize()
            ref = out.clone()
            bad, lanes, cols, dmax = 0, {}, {}, 0.0
            for it in range(a.launches):
                launch()
                if it % 50 == 49 or it == a.launches - 1:
                    pass
                sv.synchronize()
                neq = out.view(torch.int32).flatten() != ref.view(torch.int32).flatten()
                if neq.any():
                    bad += 1
                    dmax = max(dmax, float((out.flatten() - ref.flatten()).abs().max()))
                    for i in neq.nonzero().flatten().tolist():
                        if kind == 6:
                            cols[i % 32] = cols.get(i % 32, 0) + 1
                        i = i // 32 if kind == 6 else i
                        lanes[(i % 64) // 16] = lanes.get((i % 64) // 16, 0) + 1
        print('victim %-24s aggressor %s: %d of %d launches differ; differing values by lane quarter %s, by column %s, max |diff| %.3g' % (names[kind], a.aggressor if with_agg else 'none', bad, a.launches, dict(sorted(lanes.items())), dict(sorted(cols.items())), dmax))
stop.set()
th.join()

# ---- both roles inside ONE kernel (no second stream, no second queue) --------------------------------------------------------------------
lib.mix_launch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
frames = 66 * 256
xin = torch.tanh(0.3 + 0.01 * torch.randn(4 * frames + 64, device=dev))
sink = torch.empty(frames, device=dev)
for mode in (0, 1):
    for iters in (0, 200):
        spec = torch.zeros(frames, 32, device=dev)
        lib.mix_launch(mode, xin.data_ptr(), spec.data_ptr(), sink.data_ptr(), frames, 0, None)
        torch.cuda.synchronize()
        ref = spec.clone()
        bad, lanes = 0, {}
        for it in range(a.launches):
            lib.mix_launch(mode, xin.data_ptr(), spec.data_ptr(), sink.data_ptr(), frames, iters, None)
            torch.cuda.synchronize()
            neq = (spec.view(torch.int32) != ref.view(torch.int32)).any(dim=1)
            if neq.any():
                bad += 1
                for i in neq.nonzero().flatten().tolist():
                    lanes[(i % 64) // 16] = lanes.get((i % 64) // 16, 0) + 1
        print('one kernel, roles by %s, %3d MFMA iterations beside the DFT waves: %d of %d launches differ from the MFMA-free launch; rows by lane quarter %s'
              % ('wave (512-thread workgroups)' if mode == 0 else 'workgroup', iters, bad, a.launches, dict(sorted(lanes.items()))))
