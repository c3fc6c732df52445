"""BASELINE.json configs[0]: HydraVox-CV3, inference_head_num = 1, ONE 64-char utterance — 64 text ids -> 352 speech tokens (min = max token/text ratio
5.5) -> 704 mel frames -> 337 920 samples — at FULL CV3 depth, end to end, against the reference's three stages run back to back on
the CPU of the build container (tests/golden/make_golden_fullsize.py: gen_single; `llm.inference` uncached as shipped, the flow decoder driven below
its dtype wrapper, `hift.inference`).  The reference's "CPU fallback" of that config is plumbing the product refuses by design (no CPU path: the
oracle rule); what the config pins here is the SHAPE: the one-request-at-a-time loop of server/model_utils/infer_speech_model.py:612-681 at K = 1.

  fp32 mode: ids == the reference's, mel within 1e-3, f0 / source / decode stage-wise within the vocoder bounds of tests/test_gpu_cv3w.py.
  production mode (bf16 LM, reference-precision bf16 flow): teacher-forced on the reference ids — mel bound stated; ids agreement printed."""
from functools import partial

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden, state_checksum

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _scale_rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12))


@pytest.fixture(scope='module')
def single():
    from flowmirror_hydravox_amd import cv3_config
    from flowmirror_hydravox_amd import weights as W
    from oracle import hift_ref
    g = load_golden('single_cv3.npz')
    cfg = cv3_config()
    llm_sd = W.make_llm_state(cfg.llm, seed=1986, init='fan_in', with_lm_head=True)
    flow_sd = W.make_flow_state(cfg.flow, seed=1987, init='fan_in')
    hift_sd = W.make_hift_state(cfg.hift, seed=1988, init='fan_in')
    assert state_checksum(llm_sd) == str(g['llm_sha']) and state_checksum(flow_sd) == str(g['flow_sha']) and state_checksum(hift_sd) == str(g['hift_sha'])
    tables = hift_ref.make_tables(cfg.hift, seed=9)
    top_p, top_k, win, tau = g['sampling']
    return g, cfg, llm_sd, flow_sd, hift_sd, tables, dict(top_p=float(top_p), top_k=int(top_k), win_size=int(win), tau_r=float(tau))


def _pipe(single, llm_dtype, flow_dtype):
    from flowmirror_hydravox_amd.pipeline import HvxPipeline
    from flowmirror_hydravox_amd.sampling import ras_sampling
    g, cfg, llm_sd, flow_sd, hift_sd, tables, sampling = single
    return HvxPipeline(cfg, llm_sd, flow_sd, hift_sd, llm_dtype=llm_dtype, flow_dtype=flow_dtype, max_batch=1, max_ctx=512, max_t=768,
                       hift_tables=tables, sampling=partial(ras_sampling, **sampling), inference_head_num=1)


def _utt(g):
    from flowmirror_hydravox_amd.pipeline import Utterance
    return Utterance(text=torch.from_numpy(g['text']), seed=int(g['seed']), embedding=torch.from_numpy(g['emb'][0]))


def _wave_err(w, g):
    w = np.asarray(w, dtype=np.float32).reshape(-1)
    assert w.size == g['wav_f16'].size
    return max(np.abs(w[::16] - g['wav_s16']).max(), np.abs(w[:32768] - g['wav_head']).max(), np.abs(w[-32768:] - g['wav_tail']).max()), \
        np.abs(w - g['wav_f16'].astype(np.float32)).max()


def test_configs0_one_64_char_utterance_head_num_1_fp32_vs_reference(single):
    from oracle import hift_ref
    g, cfg = single[0], single[1]
    pipe = _pipe(single, torch.float32, torch.float32)
    u = _utt(g)
    wavs, st = pipe.synthesize([u], max_token_text_ratio=5.5, min_token_text_ratio=5.5)
    want = g['tokens'].tolist()
    assert len(want) == 352 and st.per_utt_tokens == [352]
    assert list(st.token_ids[0]) == want                                               # ids bit-exact, K = 1, 352 steps at contexts 66 .. 417
    # the reference's own serving call (infer_speech_model.py:630-661): llm.inference as a generator of ints, then flow.inference, then hift.inference
    pipe.llm.inference_head_num = 1
    toks = [int(t) for t in pipe.llm.inference(text=u.text[None], text_len=torch.tensor([64], dtype=torch.int32), prompt_text=torch.zeros(1, 0, dtype=torch.int32),
                                               prompt_text_len=torch.tensor([0], dtype=torch.int32), prompt_speech_token=None,
                                               prompt_speech_token_len=torch.tensor([0], dtype=torch.int32), embedding=torch.zeros(0, 192),
                                               max_token_text_ratio=5.5, min_token_text_ratio=5.5, seed=int(g['seed']))]
    assert toks == want
    mel, _ = pipe.flow.inference(token=torch.tensor(toks, dtype=torch.int32, device=DEV)[None], token_len=torch.tensor([352], dtype=torch.int32),
                                 embedding=u.embedding[None].to(DEV), finalize=True)
    assert tuple(mel.shape) == (1, 80, 704)
    e_mel = _scale_rel(mel.cpu().numpy(), g['mel'])
    # vocoder stage-wise on the REFERENCE mel: f0, source from the reference f0, decode with the reference's source (recomputed by the oracle from its f0)
    mel_ref = torch.from_numpy(g['mel'])
    f0 = pipe.hift.f0(mel_ref[0]).cpu()
    d_f0 = float((f0 - torch.from_numpy(g['f0'][0])).abs().max())
    f0_ref = torch.from_numpy(g['f0'])
    s_ref = hift_ref.source_module(F.interpolate(f0_ref[:, None], scale_factor=float(cfg.hift.upsample_total), mode='nearest').transpose(1, 2),
                                   single[4], cfg.hift, single[5]).reshape(-1)
    d_or = max(np.abs(s_ref.numpy()[::16] - g['src_s16']).max(), np.abs(s_ref.numpy()[:32768] - g['src_head']).max())
    s = pipe.hift.source(f0_ref[0]).cpu().numpy()
    d_s = max(np.abs(s[::16] - g['src_s16']).max(), np.abs(s[:32768] - g['src_head']).max(), np.abs(s[-32768:] - g['src_tail']).max())
    wav = pipe.hift.decode(mel_ref[0], s_ref).cpu().numpy()
    d_w, d_w16 = _wave_err(wav, g)
    rel_w = float(np.linalg.norm(wav[::16] - g['wav_s16']) / np.linalg.norm(g['wav_s16']))
    w_e2e = wavs[0].cpu().numpy().reshape(-1)
    d_e2e, _ = _wave_err(w_e2e, g)
    d_first = float(np.abs(w_e2e[:24000] - g['wav_head'][:24000]).max())
    print('configs[0] (K = 1, 64 chars -> 352 tokens -> 704 frames), fp32 mode vs the REFERENCE: ids equal; mel %.2e of its scale; f0 %.2e Hz; source %.2e '
          '(oracle source vs stored samples %.1e); decode(reference mel, reference source) max |d| %.2e, relative L2 %.2e; end to end: first second %.2e, whole utterance %.2e '
          '(phase drift of an ill-conditioned fp32 phase accumulation: tests/test_gpu_refpin.py)' % (e_mel, d_f0, d_s, d_or, d_w, rel_w, d_first, d_e2e))
    assert e_mel < 1e-3, e_mel                                                          # measured 1.0e-6
    assert d_f0 < 1e-3 and d_or < 1e-6 and d_s < 1e-6, (d_f0, d_or, d_s)                # measured 4.8e-4 Hz, 3e-8, 3e-8
    assert rel_w < 3e-4 and d_w < 7e-4 and d_w16 < 1e-3, (rel_w, d_w, d_w16)            # measured 1.5e-4 / 3.3e-4 (split-bf16 convolutions)
    # End to end (own mel, own f0): held to the REFERENCE's own self-distance under last-bit perturbations of its f0 on this very mel (single_cv3_cond.npz,
    # make_golden_fullsize.py single_cond; generator.py:254-260 is ill-conditioned in f0): per second <= 2x the running maximum of that band (+ 1e-3: the fp16 copy)
    gc = load_golden('single_cv3_cond.npz')
    cases = ['noise_a', 'noise_b', 'ulp_up', 'ulp_down', 'f0_fp64']
    env = np.maximum.accumulate(np.max([gc[c + '_per_second'] for c in cases], axis=0))
    n = (w_e2e.size // 24000) * 24000
    prof = np.abs(w_e2e - g['wav_f16'].astype(np.float32))[:n].reshape(-1, 24000).max(axis=1)
    l2_e2e = float(np.linalg.norm(w_e2e[::16] - g['wav_s16']) / np.linalg.norm(g['wav_s16']))
    worst_l2 = max(float(gc[c + '_l2']) for c in cases)
    print('   end to end per second / (2 x the reference\'s own band): at most %.2f; L2 %.2e (the reference against itself: up to %.2e)' % (float((prof / (2 * env + 1e-3)).max()), l2_e2e, worst_l2))
    assert (prof <= 2.0 * env + 1e-3).all(), (prof / (2 * env + 1e-3)).max()
    assert l2_e2e < 2.5 * worst_l2, (l2_e2e, worst_l2)                                  # measured 1.9x (3.76e-2 against 2.02e-2)


def test_configs0_production_mode_teacher_forced_vs_reference(single):
    """bf16 LM + the flow's reference-precision bf16 mode on the same utterance: the LM's greedy agreement with the reference ids is printed (bf16 sampling
    decisions are not bit-exact by contract), the mel of the REFERENCE ids is held to the production bound, the waveform stage-wise as above."""
    g, cfg = single[0], single[1]
    pipe = _pipe(single, torch.bfloat16, torch.bfloat16)
    u = _utt(g)
    want = g['tokens'].tolist()
    wavs, st = pipe.synthesize([u], max_token_text_ratio=5.5, min_token_text_ratio=5.5)
    got = list(st.token_ids[0])
    same = 0
    while same < min(len(got), len(want)) and got[same] == want[same]:
        same += 1
    assert len(got) == 352 and wavs[0].numel() == 337920 and torch.isfinite(wavs[0]).all()
    mel, _ = pipe.flow.inference(token=torch.tensor(want, dtype=torch.int32, device=DEV)[None], token_len=torch.tensor([352], dtype=torch.int32),
                                 embedding=u.embedding[None].to(DEV), finalize=True)
    e_mel = _scale_rel(mel.cpu().numpy(), g['mel'])
    print('configs[0], production mode: the bf16 LM follows the reference ids for the first %d of 352 tokens (one flipped decision changes every later draw); '
          'mel of the reference ids %.2e of its scale' % (same, e_mel))
    assert e_mel < 3e-3, e_mel
