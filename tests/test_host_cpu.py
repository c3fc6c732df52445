"""CPU-only tests: the C-ABI library loads and exports every declared symbol, host-side logic, DP sharding and the
world_size-2 gloo gather.  No compute call touches a GPU here."""
import ctypes
import os
import re
import subprocess
import sys
from functools import partial

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return str(sk.getsockname()[1])


# ---- C ABI ---------------------------------------------------------------------------------------------------------------
def _header_symbols():
    src = open(os.path.join(ROOT, 'include', 'hvx.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(hvx_[a-z0-9_]+)\s*\(', src)))


def test_library_builds_and_exports_every_declared_symbol():
    from flowmirror_hydravox_amd import build as hvx_build, _lib
    path = hvx_build.build()
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    declared = _header_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), 'libhvx.so does not export %s' % name
    # the ctypes binding covers exactly the header
    assert sorted(_lib.SYMBOLS.keys()) == declared
    assert _lib.load().hvx_abi_version() == _lib.HVX_ABI_VERSION == 4


def test_libhvx_holds_no_packed_fp32_instruction(tmp_path):
    """docs/history/DESIGN_rounds1-4.md §8 / build.py: on MI355X a wave's v_pk_{fma,mul,add}_f32 return wrong values in lanes 48-63 now and then while a wave of
    another queue streams MFMAs on the same SIMD (tools/mfma_interference.py), so the device code of the shipped library must not contain one."""
    import shutil
    from flowmirror_hydravox_amd import build as hvx_build
    objdump = '/opt/rocm/lib/llvm/bin/llvm-objdump'
    if not os.path.exists(objdump):
        pytest.skip('llvm-objdump not present')
    so = hvx_build.build(force=False)
    shutil.copy(so, tmp_path / 'libhvx.so')
    subprocess.run([objdump, '--offloading', 'libhvx.so'], cwd=tmp_path, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    objs = sorted(f for f in os.listdir(tmp_path) if 'amdgcn' in f)
    assert objs, 'no gfx950 code object found in libhvx.so'
    n_mfma = 0
    for f in objs:
        asm = subprocess.run([objdump, '-d', f], cwd=tmp_path, check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
        hits = sorted(set(re.findall(r'v_pk_(?:fma|mul|add)_f32', asm)))
        assert not hits, (f, hits)
        n_mfma += asm.count('v_mfma_')
    assert n_mfma > 1000                   # (the scan did see the kernels)


def test_ctypes_struct_sizes_match_the_c_header(tmp_path):
    """sizeof() of every argument struct as seen by a C compiler == the ctypes mirror (catches field drift)."""
    from flowmirror_hydravox_amd import _lib
    c = tmp_path / 's.c'
    c.write_text('#include <stdio.h>\n#include "hvx.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(hvx_sample_args), '
                 'sizeof(hvx_gemm_args), sizeof(hvx_attn_args), sizeof(hvx_llm_config), sizeof(hvx_flow_config), sizeof(hvx_hift_config), sizeof(hvx_decode_args), sizeof(hvx_nd));return 0;}\n')
    exe = tmp_path / 's'
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), str(c), '-o', str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [ctypes.sizeof(t) for t in (_lib.SampleArgs, _lib.GemmArgs, _lib.AttnArgs, _lib.LLMConfig, _lib.FlowConfig, _lib.HiftConfig, _lib.DecodeArgs, _lib.NdDesc)]
    assert got == want


def test_product_path_fails_loudly_without_a_gpu():
    from flowmirror_hydravox_amd import _lib
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(_lib.HvxError):
        _lib.require_gpu()
    from flowmirror_hydravox_amd.llm import HvxLLM
    from flowmirror_hydravox_amd.config import tiny_config
    with pytest.raises(_lib.HvxError):
        HvxLLM(tiny_config().llm)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'flowmirror_hydravox_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                txt = open(os.path.join(dirpath, f), errors='ignore').read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', txt, flags=re.M), f


# ---- host logic ----------------------------------------------------------------------------------------------------------
def test_sampling_params_from_partial_like_the_worker_builds_it():
    from flowmirror_hydravox_amd.sampling import ras_sampling, sampling_params, rep_threshold, DEFAULTS
    p = sampling_params(partial(ras_sampling, top_p=0.9, top_k=10, win_size=24, tau_r=0.2))
    assert p == dict(top_p=0.9, top_k=10, win_size=24, tau_r=0.2)
    assert sampling_params(None) == DEFAULTS

    def foreign_ras(weighted_scores, decoded_tokens, sampling, top_p=0.8, top_k=25, win_size=10, tau_r=0.1):   # the reference's own function
        raise AssertionError
    assert sampling_params(partial(foreign_ras, top_k=5))['top_k'] == 5
    with pytest.raises(ValueError):
        sampling_params(partial(ras_sampling, top_k=500))
    # integer form of `rep_num >= win_size * tau_r` with Python's double product
    for win in (4, 10, 24, 32):
        for tau in (0.05, 0.1, 0.2, 0.5):
            thr = rep_threshold(win, tau)
            for rep in range(0, win + 1):
                assert (rep >= win * tau) == (rep >= thr)


def test_noise_stream_equals_reference_draws_and_rewinds_the_generator():
    from flowmirror_hydravox_amd.sampling import NoiseStream
    from oracle import sampler_ref
    a = NoiseStream(seed=5, chunk=100)
    b = sampler_ref.NoiseStream(seed=5)
    w = a.window(0, 1000).copy()
    assert np.array_equal(w, b.peek(0, 1000))
    assert np.array_equal(a.window(700, 50), b.peek(700, 50))
    # multinomial(1) == argmax(p / q) on the same stream
    g = torch.Generator().manual_seed(9)
    p = torch.rand(40, generator=torch.Generator().manual_seed(1))
    ns = NoiseStream(seed=9)
    for k in range(20):
        idx = int(p.multinomial(1, replacement=True, generator=g))
        q = torch.from_numpy(ns.window(40 * k, 40).copy())
        assert idx == int((p / q).argmax())
    # global generator: after finalize(n) the next global draw is stream position n
    torch.manual_seed(3)
    s = NoiseStream()
    ref = s.window(0, 500).copy()
    s.finalize(123)
    assert torch.empty(1).exponential_(1.0).item() == float(ref[123])


def test_euler_schedule_matches_oracle_accumulation():
    from flowmirror_hydravox_amd.flow import euler_schedule
    from oracle import flow_ref
    ts, dts = euler_schedule(10)
    seen = []

    def est(x, m, mu, t, s, c):
        seen.append(float(t[0]))
        return torch.zeros_like(x)
    flow_ref.solve_euler(torch.zeros(1, 80, 4), flow_ref.cosine_t_span(10), torch.zeros(1, 80, 4), torch.ones(1, 1, 4), torch.zeros(1, 80),
                         torch.zeros(1, 80, 4), est, 0.7)
    assert seen == ts and len(dts) == 10 and abs(sum(dts) - 1.0) < 1e-6


def test_packing_layouts():
    from flowmirror_hydravox_amd import packing
    w = torch.arange(32 * 64, dtype=torch.float32).view(32, 64)
    p = packing.pack_frag(w).view(2, 2, 4, 16, 8)            # [N/16][K/32][g][r][j]
    for nt in range(2):
        for kt in range(2):
            for g in range(4):
                for r in (0, 7, 15):
                    assert torch.equal(p[nt, kt, g, r], w[nt * 16 + r, kt * 32 + g * 8: kt * 32 + g * 8 + 8])
    wg, wu = torch.randn(32, 32), torch.randn(32, 32)
    gu = packing.pack_gate_up(wg, wu).view(4, 1, 4, 16, 8)   # tiles: gate0, up0, gate1, up1
    assert torch.equal(gu[0, 0, 0, 3], wg[3, 0:8]) and torch.equal(gu[1, 0, 0, 3], wu[3, 0:8]) and torch.equal(gu[2, 0, 1, 0], wg[16, 8:16])
    cw = packing.conv_weight(torch.randn(5, 18, 3))
    assert cw.shape == (5, 3 * 32) and torch.all(cw.view(5, 3, 32)[:, :, 18:] == 0)


def test_checkpoint_validation_is_strict():
    from flowmirror_hydravox_amd import weights as W
    from flowmirror_hydravox_amd.config import tiny_config
    c = tiny_config().hift
    sd = W.make_hift_state(c, seed=1)
    W.check_state(sd, W.hift_spec(c), 'hift')
    sd2 = dict(sd)
    sd2.pop('conv_pre.bias')
    with pytest.raises(RuntimeError, match='missing keys'):
        W.check_state(sd2, W.hift_spec(c), 'hift')
    sd3 = dict(sd)
    sd3['conv_pre.bias'] = torch.zeros(3)
    with pytest.raises(RuntimeError, match='size mismatch'):
        W.check_state(sd3, W.hift_spec(c), 'hift')
    sd4 = dict(sd, epoch=3, step=7)                           # dropped like infer_speech_model.py:80-89
    W.check_state(sd4, W.hift_spec(c), 'hift')


def test_full_size_budget_matches_the_survey():
    from flowmirror_hydravox_amd import weights as W
    from flowmirror_hydravox_amd.config import cv3_config
    c = cv3_config()
    n = lambda spec: sum(int(np.prod(s)) for _, s, _ in spec)
    per_head = n([e for e in W.llm_spec(c.llm) if e[0].startswith('mtp_block.0.')])
    assert abs(per_head - 62.39e6) < 0.05e6
    assert abs(n(W.hift_spec(c.hift)) - 20.78e6) < 0.3e6      # + weight-norm g vectors
    dit = n([e for e in W.flow_spec(c.flow) if e[0].startswith('decoder.estimator.')])
    assert abs(dit - 331.14e6) < 0.1e6


# ---- data parallel ---------------------------------------------------------------------------------------------------------
def test_shard_by_cost_balances_and_covers():
    from flowmirror_hydravox_amd.dp import shard_by_cost
    costs = [512, 64, 300, 300, 128, 500, 64, 64, 256]
    shards = shard_by_cost(costs, 4)
    assert sorted(i for s in shards for i in s) == list(range(len(costs)))
    loads = [sum(costs[i] for i in s) for s in shards]
    assert max(loads) - min(loads) <= max(costs)
    assert shard_by_cost(costs, 4) == shards                 # deterministic


_GLOO_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from flowmirror_hydravox_amd.dp import gather_waveforms, shard_by_cost
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo')
lens = [1000, 1, 777, 0, 4242]
shards = shard_by_cost([float(l) for l in lens], world)
mine = shards[rank]
wavs = [torch.full((lens[i],), float(i)) + torch.arange(lens[i]) * 1e-3 for i in mine]
got = gather_waveforms(wavs, mine, dst=0)
if rank == 0:
    assert sorted(got) == list(range(len(lens))), sorted(got)
    for i, l in enumerate(lens):
        assert got[i].numel() == l
        if l:
            assert torch.allclose(got[i], torch.full((l,), float(i)) + torch.arange(l) * 1e-3)
    print('GATHER_OK')
else:
    assert got == {}
dist.destroy_process_group()
'''


_DEAL_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from flowmirror_hydravox_amd.dp import shard_by_cost, Handoff
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo')
g = torch.Generator().manual_seed(3)
n_text = [int(v) for v in torch.randint(64, 513, (21,), generator=g)]            # configs[3]-like mixed lengths, 21 utterances: an uneven deal
def synth(gid):                                                                   # stands in for the pipeline: a waveform that depends on the global id only
    return torch.sin(torch.arange(n_text[gid] * 3, dtype=torch.float32) * (gid + 1) * 1e-2)
shards = shard_by_cost([float(v) for v in n_text], world)
assert sorted(i for s in shards for i in s) == list(range(21)) and len(shards[0]) != len(shards[1]) or world != 2
hand = Handoff(shards, 4, dst=0)
for gid in reversed(shards[rank]):                                                # completion order is not arrival order
    hand.push(gid, synth(gid))
got = hand.finish()
if rank == 0:
    assert sorted(got) == list(range(21))
    for gid in range(21):
        assert torch.equal(got[gid], synth(gid)), gid                             # == what one rank alone produces for that global id
    loads = [sum(n_text[i] for i in s) for s in shards]
    assert max(loads) - min(loads) <= max(n_text)
    print('DEAL_OK rounds', hand.rounds)
else:
    assert got == {}
dist.destroy_process_group()
'''


def test_longest_first_deal_and_handoff_rounds_world_size_2_gloo(tmp_path):
    """the global utterance list dealt longest-first across ranks; uneven shard sizes still enter the gather collective the
    same number of times on every rank; rank 0 ends up with exactly the single-rank waveforms, keyed by global id."""
    script = tmp_path / 'd.py'
    script.write_text(_DEAL_WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=_free_port(), WORLD_SIZE='2')
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert 'DEAL_OK' in outs[0]


def test_waveform_gather_world_size_2_gloo(tmp_path):
    script = tmp_path / 'w.py'
    script.write_text(_GLOO_WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=_free_port(), WORLD_SIZE='2')
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert 'GATHER_OK' in outs[0]


def test_model_manager_surface_and_load_pt_never_raises():
    from flowmirror_hydravox_amd.model_manager import HvxModelManager
    mm = HvxModelManager()
    assert mm.models is None and mm.is_loaded is False and mm.frontend is None
    r = mm.load_pt('/nonexistent/llm.pt', '/nonexistent/flow.pt')
    assert r['status'] == 'error' and isinstance(r['message'], str)
    import argparse
    if not torch.cuda.is_available():
        with pytest.raises(ValueError):
            mm.load_models(argparse.Namespace(config=None, model_dir='/tmp', bf16=True, fp16=False, cpu=True))


class _QueueFrontend:
    """stands in for the reference's CosyVoiceFrontEnd (cli/frontend.py): text -> ids, fixed speaker / prompt features"""
    def text_normalize(self, text, split=True, text_frontend=True):
        return [text] if split else text

    def _ids(self, text):
        if 'BAD' in text:
            raise ValueError('cannot tokenise %r' % text)
        return torch.tensor([[(ord(c) * 7) % 500 for c in text]], dtype=torch.int32)

    def frontend_sft(self, text, spk_id):
        return dict(text=self._ids(text), flow_embedding=torch.full((192,), 0.01 * len(spk_id)))

    def frontend_zero_shot(self, text, prompt_text, prompt, sr, zero_shot_spk_id=''):
        n = 5
        return dict(text=self._ids(text), prompt_text=self._ids(prompt_text), llm_prompt_speech_token=torch.arange(n, dtype=torch.int32)[None],
                    flow_prompt_speech_token=torch.arange(n, dtype=torch.int32)[None], prompt_speech_feat=torch.zeros(1, 2 * n, 80),
                    flow_embedding=torch.full((1, 192), 0.02))


def test_queue_task_source_epochs_keep_fifo_around_hot_swaps_and_parameter_changes():
    """worker._TaskSource (what the continuous engine polls): consecutive synthesis tasks with the same sampling parameters join one epoch as
    they arrive; a load_pt, another parameter set, an unknown type or the sentinel closes the epoch and is carried over — never overtaken by
    what arrived behind it (reference: one task at a time, server/worker.py:54-102).  A frontend error answers that request only."""
    import queue
    import types
    from flowmirror_hydravox_amd.worker import _TaskSource
    ep = dict(top_p=0.8, top_k=25, win_size=10, tau_r=0.1, inference_head_num=2)
    mm = types.SimpleNamespace(frontend=_QueueFrontend(), configs={'sample_rate': 24000}, is_loaded=True, get_available_speakers=lambda: ['s', 'tt'])
    q, results = queue.Queue(), {}
    tasks = [dict(id=1, task_type='tts', text='abc', speaker_id='s', extra_params=dict(ep)),
             dict(id=2, task_type='zero_shot', tts_text='de', prompt_text='p', prompt_audio=None, prompt_sample_rate=16000, extra_params=dict(ep, speed=1.25), seed=7),
             dict(id=3, task_type='tts', text='BAD', speaker_id='s', extra_params=dict(ep)),
             dict(id=4, task_type='load_pt', llm_pt='x', flow_pt='y'),
             dict(id=5, task_type='tts', text='fgh', speaker_id='s', extra_params=dict(ep)),
             dict(id=6, task_type='tts', text='i', speaker_id='s', extra_params=dict(ep, top_k=10)),
             None]
    for t in tasks[1:]:
        q.put(t)
    src = _TaskSource(mm, q, results, tasks[0], lambda s: s, 0)
    u1, u2 = src.poll(False), src.poll(False)
    assert u1.tag['id'] == 1 and u1.text.tolist() == [(ord(c) * 7) % 500 for c in 'abc'] and u1.prompt_text is None and u1.speed == 1.0 and u1.seed is None
    assert u2.tag['id'] == 2 and u2.prompt_speech_token.tolist() == [0, 1, 2, 3, 4] and tuple(u2.prompt_feat.shape) == (10, 80) and u2.speed == 1.25 and u2.seed == 7
    with pytest.raises(StopIteration):                    # task 3 fails in the frontend (answered at once), task 4 closes the epoch
        src.poll(False)
    assert results == {3: {'error': "cannot tokenise 'BAD'"}} and src.carry['id'] == 4 and not src.stop
    with pytest.raises(StopIteration):                    # (a closed source stays closed: nothing behind the load_pt is taken)
        src.poll(True)
    assert q.qsize() == 3
    src = _TaskSource(mm, q, results, q.get(), lambda s: s, 0)           # the worker serves the load_pt, then starts the next epoch with task 5
    assert src.poll(False).tag['id'] == 5
    with pytest.raises(StopIteration):
        src.poll(False)
    assert src.carry['id'] == 6 and not src.stop           # other sampling parameters: llm.sampling is per-model state
    src = _TaskSource(mm, q, results, src.carry, lambda s: s, 0)
    assert src.poll(False).tag['id'] == 6
    with pytest.raises(StopIteration):
        src.poll(True)
    assert src.stop and src.carry is None                  # the shutdown sentinel
    empty = _TaskSource(mm, queue.Queue(), results, tasks[0], lambda s: s, 0)
    assert empty.poll(False).tag['id'] == 1 and empty.poll(False) is None      # nothing waiting right now: not closed


def test_queue_task_source_validates_like_text_to_speech_and_never_blocks_for_ever():
    """the batched worker's `tts` requests get the checks and the default speaker of model_manager.text_to_speech
    (infer_speech_model.py:743-780) — empty text and unknown speakers fail that request only, a request without speaker_id takes the first
    available speaker; texts over 5000 characters (the reference's segmented path) close the epoch and are carried to the one-by-one path; a
    blocking poll returns None after a bounded wait (the engine re-checks its cancellation flag), and an utterance handed back with unpoll()
    opens the next epoch instead of failing with a cancelled one."""
    import queue
    import time
    import types
    from flowmirror_hydravox_amd.worker import _TaskSource, _Tracked
    ep = dict(top_p=0.8, top_k=25, win_size=10, tau_r=0.1, inference_head_num=2)
    mm = types.SimpleNamespace(frontend=_QueueFrontend(), configs={'sample_rate': 24000}, is_loaded=True, get_available_speakers=lambda: ['s', 'tt'])
    q, results = queue.Queue(), {}
    first = dict(id=1, task_type='tts', text='abc', speaker_id='', extra_params=dict(ep))
    for t in [dict(id=2, task_type='tts', text='   ', speaker_id='s', extra_params=dict(ep)),
              dict(id=3, task_type='tts', text='abc', speaker_id='nobody', extra_params=dict(ep)),
              dict(id=4, task_type='tts', text='de', speaker_id='tt', extra_params=dict(ep)),
              dict(id=5, task_type='tts', text='x' * 5001, speaker_id='s', extra_params=dict(ep)),
              dict(id=6, task_type='tts', text='fg', speaker_id='s', extra_params=dict(ep))]:
        q.put(t)
    src = _TaskSource(mm, q, results, first, lambda s: s, 0)
    u1 = src.poll(False)
    assert u1.tag['id'] == 1 and u1.speaker_id == 's'                      # no speaker in the request: the first available one
    assert float(u1.embedding[0]) == pytest.approx(0.01 * len('s'))
    u4 = src.poll(False)                                                   # tasks 2 and 3 are answered with errors on the way
    assert u4.tag['id'] == 4 and u4.speaker_id == 'tt'
    assert results[2] == {'error': 'TTS failed: text is empty'} and results[3]['error'].startswith('TTS failed: invalid speaker_id nobody')
    with pytest.raises(StopIteration):
        src.poll(False)
    assert src.carry['id'] == 5 and not src.stop                           # segmented text: an epoch boundary, nothing behind it is taken
    assert q.qsize() == 1
    # a blocking poll on an empty queue comes back (None) after a bounded wait
    idle = _TaskSource(mm, queue.Queue(), results, first, lambda s: s, 0)
    assert idle.poll(True).tag['id'] == 1
    t0 = time.time()
    assert idle.poll(True) is None and time.time() - t0 < 5 * idle.POLL_SECONDS + 1.0
    # unpoll: the utterance's task is carried into the next epoch and no longer counts as in flight
    in_flight = {}
    tr = _Tracked(_TaskSource(mm, queue.Queue(), results, first, lambda s: s, 0), in_flight)
    u = tr.poll(False)
    assert in_flight == {1: True}
    tr.unpoll(u)
    assert in_flight == {} and tr.src.carry['id'] == 1 and tr.src.closed
    with pytest.raises(StopIteration):
        tr.poll(True)


def test_stream_tts_chunk_schedule_matches_the_reference_loop():
    """stream_tts (cli/model.py:316-362 + token2wav :405-430) with stand-in models: which token prefixes reach the flow, with which
    flags, and that the pieces handed out tile the audio of the final mel cache exactly once."""
    from flowmirror_hydravox_amd.streaming import stream_tts
    calls = []

    class Flow:
        token_mel_ratio, pre_lookahead_len, device = 2, 3, 'cpu'

        def inference(self, token, token_len, embedding, finalize, prompt_token=None, prompt_token_len=None, prompt_feat=None,
                      prompt_feat_len=None, streaming=False):
            n = token.shape[1] if finalize else token.shape[1] - self.pre_lookahead_len
            calls.append((token.shape[1], bool(finalize), bool(streaming), None if prompt_token is None else prompt_token.shape[1]))
            ids = token[0, :n].float().repeat_interleave(2)                       # "mel" frame value = its token id
            return ids.view(1, 1, -1).expand(1, 80, -1).contiguous(), None

    class Hift:
        def inference(self, speech_feat, finalize=True):
            T = speech_feat.shape[2]
            wav = speech_feat[0, 0].repeat_interleave(480)                        # sample value = token id of its frame
            return (wav[:480 * T] if finalize else wav[:480 * (T - 8)]).view(1, -1), None

    hop, n_prompt, n_tok = 25, 40, 143
    toks = list(range(1000, 1000 + n_tok))
    pieces = list(stream_tts(iter(toks), Flow(), Hift(), torch.zeros(1, n_prompt, dtype=torch.int32), torch.zeros(1, 2 * n_prompt, 80),
                             torch.zeros(1, 192), token_hop_len=hop))
    pad = 50 - n_prompt                                                           # first hop is padded to a whole chunk with the prompt
    want, off = [], 0
    while n_tok - off >= (hop + pad if off == 0 else hop) + 3:
        h = hop + pad if off == 0 else hop
        want.append((off + h + 3, False, True, n_prompt))
        off += h
    want.append((n_tok, True, False, n_prompt))                                   # the reference's last call drops `stream`
    assert calls == want
    wav = torch.cat(pieces, dim=1)
    assert wav.shape == (1, 480 * 2 * n_tok)
    assert torch.equal(wav[0], torch.tensor(toks).float().repeat_interleave(960))
    # non-streaming: one call, everything at once
    calls.clear()
    one = list(stream_tts(iter(toks), Flow(), Hift(), torch.zeros(1, 0, dtype=torch.int32), torch.zeros(1, 0, 80), torch.zeros(1, 192), stream=False))
    assert calls == [(n_tok, True, False, None)] and len(one) == 1 and one[0].shape == (1, 480 * 2 * n_tok)


def test_mtp_graft_reproduces_the_reference_script():
    """checkpoint.graft_mtp_heads vs the hashes of what scripts/post_process/add_mtp_weights_to_cosyvoice3lm_ckpt.py wrote for the same
    input checkpoint (tests/golden/make_golden.py::gen_graft runs the script itself): keys, shapes, dtypes and every byte."""
    import hashlib
    from conftest import load_golden
    from flowmirror_hydravox_amd.checkpoint import graft_mtp_heads
    g = load_golden('graft_tiny.npz')
    for ci in range(int(g['n_cases'])):
        p = 'c%d_' % ci
        hidden, vocab = int(g[p + 'hidden']), int(g[p + 'vocab'])
        gen = torch.Generator().manual_seed(int(g[p + 'input_seed']))
        sd = {'speech_embedding.weight': torch.randn(vocab, hidden, generator=gen), 'llm_decoder.weight': torch.randn(vocab, hidden, generator=gen),
              'llm_decoder.bias': torch.randn(vocab, generator=gen), 'step': torch.tensor(5)}
        if int(g[p + 'container']):
            sd['mtp_block.0.input_layernorm.weight'] = torch.full((hidden,), 0.5)
        before = {k: v.clone() for k, v in sd.items()}
        out, added = graft_mtp_heads(sd, head_num=int(g[p + 'head_num']), mtp_head_num=int(g[p + 'mtp_head_num']), seed=int(g[p + 'seed']))
        assert added == int(g[p + 'added'])
        assert sorted(out) == list(g[p + 'keys'])
        for k, sha, shp, dt in zip(g[p + 'keys'], g[p + 'sha'], g[p + 'shapes'], g[p + 'dtypes']):
            t = out[str(k)]
            assert str(tuple(t.shape)) == str(shp) and str(t.dtype) == str(dt), k
            raw = t.contiguous().view(torch.uint8).numpy().tobytes() if t.dim() else t.numpy().tobytes()
            assert hashlib.sha256(raw).hexdigest() == str(sha), k
        assert all(torch.equal(sd[k], before[k]) for k in before)                # the input dict is not modified
    with pytest.raises(KeyError):
        graft_mtp_heads({'llm_decoder.weight': torch.zeros(4, 4)})
    with pytest.raises(ValueError):
        graft_mtp_heads({'speech_embedding.weight': torch.zeros(100, 8), 'llm_decoder.weight': torch.zeros(100, 8)})


def test_packed_weight_cache_roundtrip_and_validation(tmp_path):
    """save_packed / load_packed / load_or_pack with a stand-in model: tensors and dtypes survive, and a file written for another
    configuration, dtype, layout or source checkpoint is refused (then rebuilt by load_or_pack)."""
    import dataclasses
    from flowmirror_hydravox_amd import checkpoint as ck

    @dataclasses.dataclass
    class Cfg:
        dim: int = 8
        depth: int = 2

    class Model:
        def __init__(self, cfg, dtype=torch.bfloat16):
            self.cfg, self.dtype, self.max_t, self._weights, self.loads = cfg, dtype, 64, None, 0

        def load_state_dict(self, sd):
            self.loads += 1
            return self.load_packed([sd['a'].to(self.dtype) * 2, sd['b'].float()])

        def load_packed(self, ws):
            self._weights = list(ws)
            return self

    pt = tmp_path / 'm.pt'
    torch.save({'a': torch.arange(12.).view(3, 4), 'b': torch.ones(5)}, pt)
    loader = lambda p: torch.load(p)
    m = Model(Cfg())
    assert ck.load_or_pack(m, str(pt), str(tmp_path / 'cache'), loader) == 'packed' and m.loads == 1
    m2 = Model(Cfg())
    assert ck.load_or_pack(m2, str(pt), str(tmp_path / 'cache'), loader) == 'cache' and m2.loads == 0
    assert all(torch.equal(x, y) and x.dtype == y.dtype for x, y in zip(m._weights, m2._weights))
    cache = next((tmp_path / 'cache').iterdir())
    meta = ck.read_packed_meta(str(cache))
    assert meta['kind'] == 'Model' and meta['layout'] == ck.PACK_LAYOUT and meta['n'] == '2'
    for other in (Model(Cfg(depth=3)), Model(Cfg(), dtype=torch.float32)):
        with pytest.raises(ValueError):
            ck.load_packed(other, str(cache))
    with pytest.raises(ValueError):
        ck.load_packed(Model(Cfg()), str(cache), source='0:0:other')
    # a changed checkpoint invalidates the cache
    torch.save({'a': torch.zeros(3, 4), 'b': torch.ones(5)}, pt)
    m3 = Model(Cfg())
    assert ck.load_or_pack(m3, str(pt), str(tmp_path / 'cache'), loader) == 'packed' and float(m3._weights[0].sum()) == 0.0
    with pytest.raises(ValueError):
        ck.save_packed(Model(Cfg()), str(tmp_path / 'x.hvxpack'))


# ---- long-text segmentation (infer_speech_model.py:263-452, 782-800) ------------------------------------------------------
def test_text_segmentation_equals_the_reference_functions():
    """split_text_by_punctuation / merge_short_segments against outputs of the reference's own functions (tests/golden/segmentation.json,
    minted by tests/golden/make_golden_text.py): empty and punctuation-only texts, Han and Latin runs, no punctuation at all, 10..5200 chars."""
    import json
    from flowmirror_hydravox_amd.model_manager import merge_short_segments, split_text_by_punctuation
    g = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'segmentation.json'), encoding='utf-8'))
    assert len(g['cases']) >= 80
    for c in g['cases']:
        seg = split_text_by_punctuation(c['text'], c['max_length'], c['min_length'])
        assert seg == c['split'], (c['text'][:40], c['max_length'], c['min_length'])
        assert merge_short_segments(seg, c['min_length']) == c['merged']
        assert ''.join(seg) == c['text'] or c['text'] == ''


class _SegFrontend:
    spk2info = {'spk_a': {}, 'spk_b': {}}

    def text_normalize(self, text, split=True, text_frontend=True):
        return [text] if split else text


def test_text_to_speech_takes_the_segmented_path_above_5000_chars(monkeypatch):
    """> 5000 characters: pieces of <= ~30 characters, every piece plain speaker TTS (last_prompt=False), 50-150 ms of silence between
    pieces, segments_info = {total_segments, segments}; shorter texts: one inference_tts call and segments_info None (:782-800)."""
    from flowmirror_hydravox_amd import model_manager as mm
    calls = []

    def fake_tts(manager, text, spk, speed=1.0):
        calls.append((text, spk, speed))
        return torch.full((1, 240 * len(text)), float(len(calls)))
    monkeypatch.setattr(mm, 'inference_tts', fake_tts)
    man = mm.HvxModelManager()
    man.is_loaded, man.frontend, man.configs = True, _SegFrontend(), {'sample_rate': 24000}
    short = mm.text_to_speech(man, 'hello, world.', 'spk_b', speed=1.25)
    assert short['segments_info'] is None and calls == [('hello, world.', 'spk_b', 1.25)] and short['speaker_id'] == 'spk_b'
    calls.clear()
    text = ('一二三四五六七八九十，' * 3 + 'abcdefghij klmnopqrst. ') * 110
    assert len(text) > 5000
    out = mm.text_to_speech(man, text, '', speed=1.0)                    # no speaker id: the first available one
    segs = mm.merge_short_segments(mm.split_text_by_punctuation(text, 30, 10), 10)
    assert out['speaker_id'] == 'spk_a' and out['segments_info'] == {'total_segments': len(segs), 'segments': segs}
    assert [c[0] for c in calls] == segs and len(segs) > 100
    spoken = sum(240 * len(s) for s in segs)
    gaps = out['output_audio'].shape[-1] - spoken
    assert (len(segs) - 1) * 1200 <= gaps <= (len(segs) - 1) * 3600      # 50..150 ms at 24 kHz per joint
    assert out['duration'] == out['output_audio'].shape[-1] / 24000
    with pytest.raises(ValueError):
        mm.text_to_speech(man, 'x', 'nobody')
    with pytest.raises(ValueError):
        mm.text_to_speech(man, '   ', 'spk_a')


# ---- hydravox.yaml (infer_speech_model.py:59-62) ---------------------------------------------------------------------------
_YAML = """
# fixed params
sample_rate: 24000
llm_input_size: 128
llm_output_size: 128
spk_embed_dim: 192
qwen_pretrain_path: ''
token_frame_rate: 25
token_mel_ratio: 2
chunk_size: 25
base: 96

llm: !new:cosyvoice.llm.llm_multi_head_v3.CosyVoice3LM
    llm_input_size: !ref <llm_input_size>
    llm_output_size: !ref <llm_output_size>
    speech_token_size: 6561
    length_normalized_loss: True
    lsm_weight: 0
    mix_ratio: [5, 15]
    head_num: 3
    inference_head_num: 2
    mtp_head_num: 2
    llm: !new:cosyvoice.llm.llm_multi_head_v3.Qwen2Encoder
        pretrain_path: !ref <qwen_pretrain_path>
    sampling: !name:cosyvoice.utils.common.ras_sampling
        top_p: 0.8
        top_k: 25
        win_size: 10
        tau_r: 0.1

flow: !new:cosyvoice.flow.flow.CausalMaskedDiffWithDiT
    input_size: 80
    output_size: 80
    spk_embed_dim: !ref <spk_embed_dim>
    output_type: 'mel'
    vocab_size: 6561
    input_frame_rate: !ref <token_frame_rate>
    only_mask_loss: True
    token_mel_ratio: !ref <token_mel_ratio>
    pre_lookahead_len: 3
    pre_lookahead_layer: !new:cosyvoice.transformer.upsample_encoder.PreLookaheadLayer
        in_channels: 80
        channels: 256
        pre_lookahead_len: 3
    decoder: !new:cosyvoice.flow.flow_matching.CausalConditionalCFM
        in_channels: 240
        n_spks: 1
        spk_emb_dim: 80
        cfm_params: !new:omegaconf.DictConfig
            content:
                sigma_min: 1e-06
                solver: 'euler'
                t_scheduler: 'cosine'
                training_cfg_rate: 0.2
                inference_cfg_rate: 0.5
                reg_loss_type: 'l1'
        estimator: !new:cosyvoice.flow.DiT.dit.DiT
            dim: 512
            depth: 4
            heads: 8
            dim_head: 64
            ff_mult: 2
            mel_dim: 80
            mu_dim: 80
            spk_dim: 80
            out_channels: 80
            static_chunk_size: !ref <chunk_size> * <token_mel_ratio>
            num_decoding_left_chunks: -1

hift: !new:cosyvoice.hifigan.generator.CausalHiFTGenerator
    in_channels: 80
    base_channels: !ref <base>
    nb_harmonics: 8
    sampling_rate: !ref <sample_rate>
    nsf_alpha: 0.1
    nsf_sigma: 0.003
    nsf_voiced_threshold: 10
    upsample_rates: [8, 5, 3]
    upsample_kernel_sizes: [16, 11, 7]
    istft_params:
        n_fft: 16
        hop_len: 4
    resblock_kernel_sizes: [3, 7]
    resblock_dilation_sizes: [[1, 3, 5], [1, 3, 5]]
    source_resblock_kernel_sizes: [7, 7, 11]
    source_resblock_dilation_sizes: [[1, 3, 5], [1, 3, 5], [1, 3, 5]]
    lrelu_slope: 0.1
    audio_limit: 0.99
    conv_pre_look_right: 4
    f0_predictor: !new:cosyvoice.hifigan.f0_predictor.CausalConvRNNF0Predictor
        num_class: 1
        in_channels: 80
        cond_channels: 64
"""


def test_hydravox_yaml_dimensions(tmp_path):
    """yaml_config reads the plain numbers of a HyperPyYAML model description: tagged nodes, `!ref <key>` and `!ref <a> * <b>`,
    the DictConfig wrapper of cfm_params, Qwen2 sizes from CosyVoice-BlankEN/config.json; load_models' config lookup picks it up."""
    import json
    from flowmirror_hydravox_amd.model_manager import _load_config
    from flowmirror_hydravox_amd.yaml_config import config_from_model_dir, load_hyperpyyaml_plain
    d = tmp_path / 'model'
    (d / 'CosyVoice-BlankEN').mkdir(parents=True)
    (d / 'hydravox.yaml').write_text(_YAML)
    (d / 'CosyVoice-BlankEN' / 'config.json').write_text(json.dumps(dict(hidden_size=128, num_hidden_layers=3, num_attention_heads=2, num_key_value_heads=1,
                                                                         intermediate_size=320, rope_theta=1000000.0, rms_norm_eps=1e-6, vocab_size=151936)))
    doc = load_hyperpyyaml_plain(_YAML)
    assert doc['flow']['decoder']['estimator']['static_chunk_size'] == 50 and doc['hift']['base_channels'] == 96
    assert doc['llm']['__tag__'].endswith('CosyVoice3LM')
    cfg, extras = config_from_model_dir(str(d))
    assert (cfg.llm.hidden, cfg.llm.layers, cfg.llm.q_heads, cfg.llm.kv_heads, cfg.llm.inter) == (128, 3, 2, 1, 320)
    assert (cfg.llm.head_num, cfg.llm.mtp_heads, cfg.llm.speech_tokens) == (3, 2, 6561)
    assert extras == {'sampling': {'top_p': 0.8, 'top_k': 25, 'win_size': 10, 'tau_r': 0.1}, 'inference_head_num': 2}
    f = cfg.flow
    assert (f.dim, f.depth, f.heads, f.ff_mult, f.static_chunk_size, f.cfg_rate, f.pre_lookahead_channels, f.vocab) == (512, 4, 8, 2, 50, 0.5, 256, 6561)
    h = cfg.hift
    assert (h.base_channels, h.f0_channels, h.resblock_kernel_sizes, h.upsample_rates, h.n_fft, h.hop, h.sampling_rate) == (96, 64, [3, 7], [8, 5, 3], 16, 4, 24000)
    assert h.upsample_total == 480 and cfg.sample_rate == 24000
    got, ex2 = _load_config(str(d))
    assert got == cfg and ex2 == extras
    # an explicit hvx_config.json still wins; an unsupported head size is refused by name
    (d / 'hydravox.yaml').write_text(_YAML.replace('dim_head: 64', 'dim_head: 32'))
    with pytest.raises(ValueError, match='dim_head'):
        config_from_model_dir(str(d))
    # no yaml: None (the caller falls back to the preset)
    assert config_from_model_dir(str(tmp_path)) is None


@pytest.mark.skipif(not os.path.isdir('/root/reference'), reason='reference tree not present (GPU box)')
def test_yaml_fixture_uses_only_real_constructor_arguments():
    """every key the test yaml gives a `!new:` node is an argument of that reference class's __init__ (checked on the sources' AST:
    the modules themselves need packages that are absent here)"""
    import ast
    from flowmirror_hydravox_amd.yaml_config import load_hyperpyyaml_plain
    base = '/root/reference/server/model_utils/'
    doc = load_hyperpyyaml_plain(_YAML)

    def init_args(tag):
        mod, cls = tag.split(':', 1)[1].rsplit('.', 1)
        path = base + mod.replace('.', '/') + '.py'
        tree = ast.parse(open(path, encoding='utf-8').read())
        c = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls)
        f = next(n for n in c.body if isinstance(n, ast.FunctionDef) and n.name == '__init__')
        return {a.arg for a in f.args.args + f.args.kwonlyargs} - {'self'}

    def walk(node, seen):
        if isinstance(node, dict):
            tag = node.get('__tag__', '')
            if tag.startswith('!new:cosyvoice.'):
                args = init_args(tag)
                for k in node:
                    assert k == '__tag__' or k in args, (tag, k)
                seen.append(tag)
            for v in node.values():
                walk(v, seen)
    seen = []
    walk(doc, seen)
    assert len(seen) >= 7


def test_frontend_gemm_bases_reproduce_the_oracle_features():
    """the folded GEMM operands the device path uses (packing.whisper_bases / kaldi_fbank_bases: window, DC removal, pre-emphasis and FFT
    zero-padding inside ONE basis matrix; 400-sample frames as 13 rows of 32) evaluated in float64 on the host == the step-by-step
    torch.stft / torch.fft restatement of oracle/frontend_ref.py"""
    import torch.nn.functional as F
    from flowmirror_hydravox_amd.packing import kaldi_fbank_bases, mel_filterbank, whisper_bases
    from oracle import frontend_ref as R
    g = torch.Generator().manual_seed(0)
    x = torch.randn(16000 * 2, generator=g) * 0.1 + 0.05

    def framed(sig, n):
        sig = F.pad(sig, (0, 416))
        return torch.stack([sig[i * 160: i * 160 + 416] for i in range(n)]).double()
    ana, mel = whisper_bases(128)
    xp = F.pad(x[None, None], (200, 200), mode='reflect')[0, 0]
    spec = framed(xp, len(x) // 160) @ ana.double().t()
    m = torch.clamp((spec[:, :201] ** 2 + spec[:, 201:] ** 2) @ mel[:, :201].double().t(), min=1e-10).log10()
    m = (torch.maximum(m, m.max() - 8.0) + 4.0) / 4.0
    assert (m.t().float() - R.whisper_log_mel(x, mel_filterbank(16000, 400, 128, 0.0, 8000.0))).abs().max().item() < 1e-4
    ana, mel = kaldi_fbank_bases(80)
    spec = framed(x, 1 + (len(x) - 400) // 160) @ ana.double().t()
    f = torch.clamp((spec[:, :257] ** 2 + spec[:, 257:] ** 2) @ mel[:, :257].double().t(), min=1.1920929e-07).log()
    assert ((f - f.mean(0, keepdim=True)).float() - R.kaldi_fbank(x[None])).abs().max().item() < 5e-4
    # the kaldi mel banks: 80 triangles, unit peak, zero outside [20 Hz, Nyquist)
    banks = R.kaldi_mel_banks()
    assert banks.shape == (80, 256) and float(banks.max()) <= 1.0 and float(banks[:, 0].max()) == 0.0 and (banks.sum(1) > 0).all()


def test_e4m3_power_of_two_quantiser_and_fragment_orders():
    """packing.quantize_e4m3_pow2 (fp8 head weights): codes decode by the OCP e4m3 definition (restated here: bias 7, subnormals at
    exponent 0, no infinities, 0x7f / 0xff = NaN) to what the cast produced, one power-of-two scale per row, |code| <= 448 reached on the row's
    largest element, code x scale exact in bf16, rounding error within half an e4m3 step; pack_frag_fp8 holds the two k-steps of pack_frag's
    fragments lane by lane, and frag_fp8_to_frag rebuilds the bf16 tensor a packed cache leaves out."""
    import torch
    from flowmirror_hydravox_amd.packing import frag_fp8_to_frag, pack_frag, pack_frag_fp8, quantize_e4m3_pow2

    def decode(code):
        s, e, m = -1.0 if code & 0x80 else 1.0, (code >> 3) & 0xf, code & 7
        if e == 0xf and m == 7:
            return float('nan')
        return s * (m / 8.0) * 2.0 ** -6 if e == 0 else s * (1 + m / 8.0) * 2.0 ** (e - 7)
    table = torch.tensor([decode(c) for c in range(256)], dtype=torch.float32)
    assert float(table[0x7e]) == 448.0 and float(table[0x01]) == 2.0 ** -9
    g = torch.Generator().manual_seed(0)
    w = torch.randn(64, 256, generator=g) * torch.logspace(-4, 1, 64)[:, None]
    w[5] = 0
    codes, scale, deq = quantize_e4m3_pow2(w)
    assert codes.dtype == torch.uint8 and codes.shape == w.shape and scale.shape == (64,)
    assert torch.equal(table[codes.long()] * scale[:, None], deq)
    assert torch.equal(torch.log2(scale), torch.log2(scale).round())
    assert torch.equal(deq.to(torch.bfloat16).float(), deq)
    amax_code = table[codes.long()].abs().amax(1)
    live = w.abs().amax(1) > 0
    assert bool((amax_code[live] >= 224).all()) and bool((amax_code <= 448).all())             # the smallest power of two that fits (224.x rounds to 224)
    step = torch.where(w.abs() / scale[:, None] >= 2.0 ** -6, 2.0 ** (torch.floor(torch.log2(w.abs() / scale[:, None]).clamp_min(-6)) - 3), torch.tensor(2.0 ** -9))
    assert bool(((deq - w).abs() <= 0.5 * step * scale[:, None] * (1 + 1e-6)).all())
    assert torch.equal(deq[5], torch.zeros(256))
    # fragment orders
    pf = pack_frag(deq).view(4, 8, 4, 16, 8)                                                   # [tile][k-step][g][r][8]
    p8 = pack_frag_fp8(codes)
    v8 = table[p8.long()].view(4, 4, 4, 16, 2, 8) * scale.view(4, 1, 1, 16, 1, 1)              # [tile][double step][g][r][half][8]
    for half in range(2):
        assert torch.equal(v8[..., half, :], pf[:, half::2])
    assert torch.equal(frag_fp8_to_frag(p8, scale, torch.bfloat16), pack_frag(deq).to(torch.bfloat16))


def test_onnx_auto_pad_and_opset_defaults_are_honoured_not_ignored():
    """a Conv / AveragePool exported with auto_pad = SAME_UPPER / SAME_LOWER / VALID must get those pads (host logic of the executor and of
    the oracle, against torch), an unknown mode is refused, and Softmax without an axis means axis 1 before opset 13."""
    import numpy as np
    import torch
    import torch.nn.functional as F
    from flowmirror_hydravox_amd import onnx_graph as og
    from oracle import onnx_ref
    assert og._resolve_pads({'auto_pad': b'SAME_UPPER'}, [23], [4], [2], [1]) == [1, 2]
    assert og._resolve_pads({'auto_pad': b'SAME_LOWER'}, [23], [4], [2], [1]) == [2, 1]
    assert og._resolve_pads({'auto_pad': 'VALID', 'pads': [3, 3]}, [23], [4], [2], [1]) == [0, 0]
    assert og._resolve_pads({'auto_pad': b'NOTSET', 'pads': [3, 5]}, [23], [4], [2], [1]) == [3, 5]
    assert og._resolve_pads({}, [11, 9], [3, 3], [1, 1], [1, 1]) == [0, 0, 0, 0]
    assert og._resolve_pads({'auto_pad': b'SAME_UPPER'}, [11, 9], [3, 5], [2, 1], [1, 2]) == [1, 4, 1, 4]
    with pytest.raises(NotImplementedError):
        og._resolve_pads({'auto_pad': b'SAME'}, [23], [4], [2], [1])
    rng = np.random.default_rng(2)
    x, w = rng.standard_normal((2, 4, 23)).astype(np.float32), rng.standard_normal((6, 4, 4)).astype(np.float32)
    for mode, pad in ((b'SAME_UPPER', (1, 2)), (b'SAME_LOWER', (2, 1)), (b'VALID', (0, 0))):
        nd = og.Node('Conv', ['x', 'w'], ['y'], dict(kernel_shape=[4], strides=[2], auto_pad=mode))
        want = F.conv1d(F.pad(torch.from_numpy(x), pad), torch.from_numpy(w), stride=2).numpy()
        np.testing.assert_allclose(onnx_ref._node(nd, [x, w], 17), want, rtol=1e-4, atol=1e-4)
    xs = rng.standard_normal((3, 5, 7)).astype(np.float32)
    want = torch.softmax(torch.from_numpy(xs).reshape(3, -1), 1).reshape(xs.shape).numpy()
    np.testing.assert_allclose(onnx_ref._node(og.Node('Softmax', ['x'], ['y'], {}), [xs], 11), want, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(onnx_ref._node(og.Node('Softmax', ['x'], ['y'], {}), [xs], 13), torch.softmax(torch.from_numpy(xs), -1).numpy(), rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize('config,world,steps,B', [('tts', 2, 3, 4), ('zero_shot', 2, 3, 4), ('zero_shot', 8, 1, 8)])
def test_bench_rank_logic_world_size_2_gloo_with_a_stub_pipeline(config, world, steps, B):
    """everything `bench.py --gpus N` does AROUND the pipeline — WORLD_SIZE / RANK from the environment, the longest-first deal of the
    global utterance list (uneven with mixed lengths), one continuous job per rank, Handoff rounds of one step's worth of waveforms to rank 0 inside the
    timed region, barrier + max-over-ranks of the wall time, sum of the token counts, ONE JSON line from rank 0 — at world size 2 and, for BASELINE configs[3]'s
    own geometry (64 mixed-length zero-shot utterances over 8 ranks, 8 per rank and step), at world size 8 on gloo, with tools/bench_stub.py standing in for the GPU
    pipeline.  Rank 0 must have received every global utterance, bit-equal to what one rank alone makes; the deal balances TEXT, not counts."""
    import json
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    from bench_stub import stub_wave
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=_free_port(), WORLD_SIZE=str(world), HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES='', OMP_NUM_THREADS='1')
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(world), '--steps', str(steps), '--warmup', '1', '--batch', str(B), '--tiny', '--stub-pipeline', '--config', config]
    procs = [subprocess.Popen(cmd, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-2000:] for o in outs]
    for r in range(1, world):
        assert not [ln for ln in outs[r][0].splitlines() if ln.startswith('{')]      # only rank 0 prints a line
    lines = [ln for ln in outs[0][0].splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    n_global = steps * B * world
    assert d['n_gpus'] == world and d['steps'] == steps and d['scaling'] == 'weak' and d['config']['parallelism'] == 'utterance-dp%d' % world and d['config']['global_batch'] == B * world
    assert d['rccl_ranks'] == world and d['collective_backend'] == 'gloo'               # (the stub's group; on GPUs: "nccl (= RCCL on ROCm)")
    assert 'STUB' in d['config']['workload'] and d['value'] > 0 and d['ms_per_step'] > 0
    st = d['stub']
    assert st['received_on_rank0'] == n_global and sum(st['shard_sizes']) == n_global
    assert st['handoff_rounds'] == -(-max(st['shard_sizes']) // B)
    if config == 'zero_shot':
        # mixed lengths: the longest-first deal balances the text the ranks decode (the spread of LPT is bounded by the largest item over the mean load)
        lo, hi = min(st['shard_text']), max(st['shard_text'])
        assert (hi - lo) / (sum(st['shard_text']) / world) < (0.03 if world == 8 else 0.06), st['shard_text']
    # the checksum of everything rank 0 holds == what a single rank produces for those global ids (seed = global index)
    import torch
    from flowmirror_hydravox_amd.config import tiny_config
    from flowmirror_hydravox_amd.pipeline import synthetic_utterance

    def n_text_of(g):
        return 512 if config == 'tts' else int(torch.randint(64, 513, (1,), generator=torch.Generator().manual_seed(7_000_003 + g)))
    want = 0.0
    tokens = 0
    for gidx in range(n_global):
        u = synthetic_utterance(tiny_config(), gidx, n_text_of(gidx)) if config == 'tts' else synthetic_utterance(tiny_config(), gidx, n_text_of(gidx), n_prompt_speech=75, n_prompt_text=20)
        n = int(int(u.text.numel()) * 5.5) // 16
        tokens += n
        want += float(stub_wave(u.seed, n).double().sum())
    assert abs(st['checksum'] - want) < 1e-6 * max(1.0, abs(want)), (st['checksum'], want)
    assert abs(d['value'] * d['ms_per_step'] * steps / 1e3 - tokens) < 0.01 * tokens      # value = ALL ranks' tokens / max-over-ranks wall time


def test_onnx_covered_set_is_what_the_tests_reach():
    """N2: the device ONNX executor refuses what it was never compared on.  onnx_graph.COVERED must equal, operator by operator and attribute by attribute, what
    the synthetic frontend graphs and the one-node cases of tests/onnx_synth.py reach (the GPU tests run exactly those against oracle/onnx_ref.py); every one-node
    case is evaluable by the oracle; `uncovered` names operators and attributes outside the table (OnnxRunner raises on them before touching the device)."""
    import onnx_synth
    from flowmirror_hydravox_amd import onnx_graph as og
    from oracle import onnx_ref
    reach = {k: tuple(sorted(v)) for k, v in onnx_synth.covered_set().items()}
    assert reach == {k: tuple(sorted(v)) for k, v in og.COVERED.items()}
    assert set(og.COVERED) <= set(og.OPERATORS)
    for op, ins, at in onnx_synth.per_operator_cases():
        out = onnx_ref._node(og.Node(op, ['i%d' % i for i in range(len(ins))], ['o'], at), ins, 17)
        assert out is not None, op
    assert og.uncovered(onnx_synth.campplus_like()) == [] and og.uncovered(onnx_synth.tokenizer_like(with_length=True)) == []
    g = og.Graph([og.Node('Conv', ['x', 'w'], ['y'], {'kernel_shape': [3], 'foo': 1}, name='n1'), og.Node('LSTM', ['y'], ['z'], {}, name='n2'),
                  og.Node('Softmax', ['z'], ['o'], {'axis': -1}, name='n3')], {}, ['x'], ['o'])
    assert og.uncovered(g) == [('n1', 'Conv', 'foo'), ('n2', 'LSTM', None)]
    assert np.array_equal(onnx_ref._node(og.Node('Mod', ['a', 'b'], ['o'], {}), [np.asarray([7, -7, 9]), np.asarray([3, 3, 4])], 17), [1, 2, 1])


def test_library_options_are_one_switchboard_and_lab_ones_are_refused():
    """hvx_set_option / hvx_get_option / hvx_option_name (include/hvx.h; csrc/hvx_options.h): no environment variable is read anywhere in csrc/, unknown keys and
    out-of-range values fail with a message, lab options cannot be set in the shipped library, and the shipped library is not a lab build."""
    from flowmirror_hydravox_amd import _lib
    lib = _lib.load()
    assert int(lib.hvx_is_lab_build()) == 0 and lib.hvx_build_flags() == b''
    opts = _lib.options()
    assert {'dec_gemm', 'conv64_resident', 'x3p8', 'att_chunk', 'att_waves', 'gemm_big_gw', 'attn_lab'} <= set(opts)
    assert all(v == d for v, d, _ in opts.values())                       # nothing has been switched in this process
    assert opts['attn_lab'][2] and not opts['dec_gemm'][2]
    with _lib.option_scope(dec_gemm=0, att_chunk=512):
        assert _lib.get_option('dec_gemm') == 0 and _lib.get_option('att_chunk') == 512
    assert _lib.get_option('dec_gemm') == 1 and _lib.get_option('att_chunk') == 0
    with _lib.option_scope(attn_dit_form=16, dec_fuse_rows=16):           # the round-5 attention tile / the fused narrow-grid o_proj stay selectable for A / B runs
        assert _lib.get_option('attn_dit_form') == 16 and _lib.get_option('dec_fuse_rows') == 16
    assert _lib.get_option('attn_dit_form') == 0 and _lib.get_option('dec_fuse_rows') == 0
    for key, val, msg in (('no_such_option', 1, 'unknown option'), ('att_chunk', 100, 'multiple of 128'), ('att_waves', 5, '4 or 8'), ('attn_lab', 3, 'lab option'),
                          ('attn_dit_form', 18, '0, 16, 17 or 32'), ('attn_dit_form', 48, 'lab builds only'), ('dec_fuse_rows', 300, '0..256')):
        with pytest.raises(_lib.HvxError, match=msg):
            _lib.set_option(key, val)
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'flowmirror_hydravox_amd', 'csrc')
    for f in os.listdir(csrc):
        if f.endswith(('.hip', '.h')):
            assert 'getenv' not in open(os.path.join(csrc, f)).read(), f


def test_build_never_reuses_objects_of_other_flags(tmp_path):
    """objects carry a stamp of the exact compiler command; another flag set (a lab build) means rebuild, and extra flags never go into the product library"""
    from flowmirror_hydravox_amd import build as B
    src = tmp_path / 'k.hip'
    src.write_text('// x')
    obj = str(tmp_path / 'k.o')
    assert B._stale(str(src), obj, 0.0, ())                                 # no object
    open(obj, 'w').write('o')
    assert B._stale(str(src), obj, 0.0, ())                                 # no stamp: never trusted
    open(obj + '.cmd', 'w').write(' '.join(B._command(str(src), obj, ())))
    assert not B._stale(str(src), obj, 0.0, ())
    assert B._stale(str(src), obj, 0.0, ('-DHVX_LAB',))                     # other flags
    assert '-DHVX_BUILD_FLAGS="-DHVX_LAB -DHVX_LAB_GEMM_EPI=1"' in B._command(str(src), obj, ('-DHVX_LAB', '-DHVX_LAB_GEMM_EPI=1'))
    with pytest.raises(ValueError):
        B.build(extra=('-DHVX_LAB',))                                       # extra flags without a lab name: refused before anything is compiled


def test_acoustic_config_deal_of_10000_streams_over_8_ranks():
    """BASELINE configs[4] on 8 GPUs: 10 000 pre-tokenised streams of U{352..2816} tokens dealt longest-first by WORK a T + b T^2 (dp.acoustic_cost: the DiT attention
    makes a long stream worth more than its length) — every rank's work within 0.1 % of the mean although the counts differ, every stream exactly once; and what
    bench.py --config acoustic does with it at one rank (all streams, same ids as before the deal existed)."""
    from flowmirror_hydravox_amd.dp import acoustic_cost, shard_by_cost
    lens = [int(torch.randint(352, 2817, (1,), generator=torch.Generator().manual_seed(9_000_011 + i))) for i in range(10000)]
    costs = [acoustic_cost(2.0 * m) for m in lens]
    shards = shard_by_cost(costs, 8)
    assert sorted(i for s in shards for i in s) == list(range(10000))
    work = [sum(costs[i] for i in s) for s in shards]
    assert (max(work) - min(work)) / (sum(work) / 8) < 1e-3, work
    frames = [sum(2 * lens[i] for i in s) for s in shards]
    assert (max(frames) - min(frames)) / (sum(frames) / 8) < 0.02            # frames follow the work closely but not exactly (quadratic term)
    assert acoustic_cost(5632) / acoustic_cost(704) > 8.5                    # 8x the frames, ~8.9x the work
    assert shard_by_cost(costs[:64], 1) == [list(range(64))]
