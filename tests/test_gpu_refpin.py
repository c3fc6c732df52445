"""The kernels the benchmark runs, AT THE BENCHMARKED SHAPES, against the reference itself — not against another form of this library.

Fixtures: tests/golden/make_golden_fullsize.py ran the reference's own modules ONCE on the CPU of the build container at BASELINE.json configs[1]'s
sizes (22 DiT blocks at T = 5632 with a padded row; the 10-step CFG solve of a 2816-token utterance; CausalHiFTGenerator.inference at 5632 frames;
the 24-layer LM on a 3300-row prefix + its own uncached K = 2 generation from that context) and at configs[0]'s (head_num = 1, one 64-char
utterance end to end: tests/test_gpu_single.py).  Held to them here:

  * the exact-fp32 forms (north_star's parity clause: ids bit-exact, mel / waveform 1e-3),
  * the PRODUCTION forms of `python bench.py` (bf16 LM on the wide decode grid, the flow's reference-precision bf16 mode with the 64-rows-per-wave
    attention and the 256-tile Linears, the split-bf16 vocoder convolutions) with bounds <= 2x what was measured on MI355X, stated per assert.
"""
import hashlib

import numpy as np
import pytest
import torch

from conftest import load_golden, state_checksum
from test_oracle_golden import cv3w_flow_inputs

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _scale_rel(a, b):
    """max |a - b| / max |b|: the measure of tests/test_gpu_cv3d.py"""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12))


def _l2_rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def _sha(t):
    return hashlib.sha256(t.detach().contiguous().numpy().tobytes()).hexdigest()


# ---- flow: 22 DiT blocks at T = 5632 -----------------------------------------------------------------------------------------------------------
@pytest.fixture(scope='module')
def flow_sd():
    from flowmirror_hydravox_amd import cv3_config
    from flowmirror_hydravox_amd import weights as W
    c = cv3_config().flow
    sd = W.make_flow_state(c, seed=1987, init='fan_in')
    return c, sd, state_checksum(sd)


# handle keyword arguments per mode; bounds on (max|d| / max|ref|, relative L2).  CONTRACT (north_star): 1e-3 — met by the fp32 forms (asserted at 1e-3, measured 2.5e-6).
# The production forms do NOT meet 1e-3 (2.4e-3; the reference's own fp16 deployment of this decoder sits at 1.36e-3 from its fp32 run): their bounds below are <= 2x what was
# measured on MI355X — a regression guard on today's arithmetic, printed next to the contract number so that nobody reads them as the contract.
_EST_KW = {'fp32': dict(dtype=torch.float32), 'production': dict(dtype=torch.bfloat16),
           'plain-bf16': dict(dtype=torch.bfloat16, f16_linears=False, f32_small=False)}
#   measured: fp32 2.5e-6 / 2.1e-6; production 2.4e-3 / 2.35e-3; plain bf16 6.6e-3 / 5.9e-3
_EST_BOUNDS = {'production': (4.5e-3, 4.5e-3), 'plain-bf16': (1.2e-2, 1.1e-2)}


@pytest.mark.parametrize('mode', ['fp32', 'production', 'plain-bf16'])
def test_dit_22_blocks_5632_frames_vs_reference(flow_sd, mode):
    """cosyvoice/flow/DiT/dit.py:145-176 at the bench shape: B = 2 (one CFG pair), T = 5632, second row padded to 5000 frames (key-padding mask at
    full size: the reference's output times its mask)."""
    from flowmirror_hydravox_amd.flow import HvxFlow
    c, sd, sd_sha = flow_sd
    g = load_golden('flow_full_est.npz')
    assert sd_sha == str(g['weight_sha'])
    T, lens = int(g['T']), g['lens'].tolist()
    x, mask, mu, spk, cond = cv3w_flow_inputs(int(g['seed']), T, lens)
    assert state_checksum(dict(x=x, mu=mu, spk=spk, cond=cond)) == str(g['in_sha'])
    flow = HvxFlow(c, sd, max_t=T + 64, **_EST_KW[mode])
    est = (flow.estimator(x, mask, mu, torch.from_numpy(g['t']), spk, cond).cpu() * mask).numpy()
    assert np.isfinite(est).all()
    e = [(_scale_rel(est[i, :, :n], g['out'][i, :, :n]), _l2_rel(est[i, :, :n], g['out'][i, :, :n])) for i, n in enumerate(lens)]
    print('DiT 22 blocks, T = %d, lens %s, %s forms vs the REFERENCE: max|d|/max|ref| %s, relative L2 %s  [contract 1e-3: %s]'
          % (T, lens, mode, ['%.2e' % a for a, _ in e], ['%.2e' % b for _, b in e], 'met' if max(a for a, _ in e) < 1e-3 else 'NOT met by this mode (regression bound only)'))
    bm, bl = (1e-3, 1e-3) if mode == 'fp32' else _EST_BOUNDS[mode]
    assert max(a for a, _ in e) < bm and max(b for _, b in e) < bl, (mode, e)
    # padded columns of the second row are zero after the mask, like the reference's
    assert np.abs(est[1, :, lens[1]:]).max() == 0.0


#   CONTRACT 1e-3 (fp32 forms: asserted); production: regression bound <= 2x measured, above the contract
#   measured: fp32 1.1e-6 / 1.1e-6; production 1.52e-3 / 1.54e-3 (alone and as entry 0 of a padded batch of 4)
_SOLVE_BOUNDS = {'fp32': (1e-3, 1e-3), 'production': (3.0e-3, 3.0e-3)}


@pytest.mark.parametrize('mode', ['fp32', 'production'])
def test_cfm_solve_of_a_512_char_utterance_vs_reference(flow_sd, mode):
    """cosyvoice/flow/flow.py:367-430 + flow_matching.py:204-228, 71-124: token embedding, pre-lookahead, x2 repeat, 10 Euler steps x CFG 2 over
    5632 frames (220 block evaluations at the bench length), through the product's own `flow.inference`."""
    from flowmirror_hydravox_amd.flow import HvxFlow
    c, sd, sd_sha = flow_sd
    g = load_golden('flow_full_solve.npz')
    assert sd_sha == str(g['weight_sha'])
    token, emb = torch.from_numpy(g['token']), torch.from_numpy(g['emb'])
    flow = HvxFlow(c, sd, dtype=torch.float32 if mode == 'fp32' else torch.bfloat16, max_t=2 * token.shape[1] + 64)
    mel, _ = flow.inference(token=token.to(DEV), token_len=torch.tensor([token.shape[1]], dtype=torch.int32), embedding=emb.to(DEV), finalize=True)
    mel = mel.cpu().numpy()
    assert mel.shape == g['mel'].shape and np.isfinite(mel).all()
    e = (_scale_rel(mel, g['mel']), _l2_rel(mel, g['mel']))
    print('10-step CFG solve, 2816 tokens -> 5632 frames, %s forms vs the REFERENCE: max|d|/max|ref| %.2e, relative L2 %.2e  [contract 1e-3: %s]'
          % (mode, e[0], e[1], 'met' if e[0] < 1e-3 else 'NOT met by this mode (regression bound only; the reference\'s own fp16 run: 1.36e-3)'))
    assert e[0] < _SOLVE_BOUNDS[mode][0] and e[1] < _SOLVE_BOUNDS[mode][1], (mode, e)
    if mode == 'production':
        # the padded batch solve of the bench (hvx_cfm_solve_batch, 4 entries x CFG 2 per estimator call): the same utterance beside three others
        gen = torch.Generator().manual_seed(5)
        others = [torch.randint(0, c.vocab, (n,), generator=gen, dtype=torch.int32).to(DEV) for n in (2816, 2500, 2050)]
        embs = [emb[0].to(DEV)] + [torch.randn(c.spk_embed_dim, generator=gen).to(DEV) for _ in others]
        mels = flow.inference_batch([token[0].to(torch.int32).to(DEV)] + others, embs)
        eb = (_scale_rel(mels[0].cpu().numpy(), g['mel']), _l2_rel(mels[0].cpu().numpy(), g['mel']))
        print('   the same utterance as entry 0 of a padded batch of 4: %.2e / %.2e' % eb)
        assert eb[0] < _SOLVE_BOUNDS[mode][0] and eb[1] < _SOLVE_BOUNDS[mode][1], eb


#   measured: fp32 1.0e-6 / 1.1e-6; production 1.66e-3 / 1.54e-3
_PROMPT_BOUNDS = {'fp32': (1e-3, 1e-3), 'production': (3.3e-3, 3.0e-3)}


@pytest.mark.parametrize('mode', ['fp32', 'production'])
def test_cfm_solve_with_a_prompt_at_full_depth_vs_reference(flow_sd, mode):
    """BASELINE configs[3]'s flow call (zero-shot: 75 prompt tokens + 150 prompt mel frames, cosyvoice/flow/flow.py:389-430) at 22 blocks and 2966 frames: prompt tokens
    prepended, the prompt mel as `cond`, the prompt frames cut from the result; alone and as a member of a mixed-length padded batch (configs[3] batches are mixed)."""
    from flowmirror_hydravox_amd.flow import HvxFlow
    c, sd, sd_sha = flow_sd
    g = load_golden('flow_full_prompt.npz')
    assert sd_sha == str(g['weight_sha'])
    token, ptoken, pfeat, emb = (torch.from_numpy(g[k]) for k in ('token', 'ptoken', 'pfeat', 'emb'))
    flow = HvxFlow(c, sd, dtype=torch.float32 if mode == 'fp32' else torch.bfloat16, max_t=2 * (token.shape[1] + ptoken.shape[1]) + 64)
    mel, _ = flow.inference(token=token.to(DEV), token_len=torch.tensor([token.shape[1]], dtype=torch.int32), embedding=emb.to(DEV), finalize=True,
                            prompt_token=ptoken.to(DEV), prompt_token_len=torch.tensor([ptoken.shape[1]], dtype=torch.int32),
                            prompt_feat=pfeat.to(DEV), prompt_feat_len=torch.tensor([pfeat.shape[1]], dtype=torch.int32))
    mel = mel.cpu().numpy()
    assert mel.shape == g['mel'].shape and np.isfinite(mel).all()
    e = (_scale_rel(mel, g['mel']), _l2_rel(mel, g['mel']))
    print('10-step CFG solve with a prompt, 1408 + 75 tokens -> 2966 frames, %s forms vs the REFERENCE: max|d|/max|ref| %.2e, relative L2 %.2e' % (mode, e[0], e[1]))
    assert e[0] < _PROMPT_BOUNDS[mode][0] and e[1] < _PROMPT_BOUNDS[mode][1], (mode, e)


# ---- vocoder at 5632 frames ----------------------------------------------------------------------------------------------------------------------
def _check_wave(tag, wav, g, p, b_s16, b_f16):
    """a waveform against the packed fixture (make_golden_fullsize.pack_wave): fp32 on every 16th sample + head / tail, fp16 everywhere"""
    w = np.asarray(wav, dtype=np.float32).reshape(-1)
    assert w.size == g[p + 'wav_f16'].size
    d16 = np.abs(w[::16] - g[p + 'wav_s16']).max()
    dh = max(np.abs(w[:32768] - g[p + 'wav_head']).max(), np.abs(w[-32768:] - g[p + 'wav_tail']).max())
    l2 = _l2_rel(w[::16], g[p + 'wav_s16'])
    dall = np.abs(w - g[p + 'wav_f16'].astype(np.float32)).max()
    print('%s: max |d| on every 16th sample %.2e, on the first / last 32 k samples %.2e, relative L2 %.2e; every sample vs the fp16 copy %.2e' % (tag, d16, dh, l2, dall))
    assert max(d16, dh) < b_s16 and dall < b_f16, (tag, d16, dh, dall)
    return l2


def _hift_production_and_exact(snippet, tmp_path, extra_args=()):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for name in ('x3', 'exact'):                  # (one process per form: each holds the full-size workspace; the form is hvx_hift_config.exact_fp32, no environment involved)
        out = str(tmp_path / (name + '.pt'))
        r = subprocess.run([sys.executable, '-c', snippet, out, root, name] + list(extra_args), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-3000:]
        outs[name] = torch.load(out)
    return outs


_HIFT_SNIPPET = r"""
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[2])
sys.path.insert(0, sys.argv[2] + '/tests')
from flowmirror_hydravox_amd import cv3_config
from flowmirror_hydravox_amd import weights as W
from flowmirror_hydravox_amd.hift import HvxHift
from oracle import hift_ref          # (tests only: the checker's source module and noise tables)
import torch.nn.functional as F
g = np.load(sys.argv[2] + '/tests/golden/hift_full.npz')
c = cv3_config().hift
sd = W.make_hift_state(c, seed=int(g['weight_seed']), init='fan_in')
tables = hift_ref.make_tables(c, seed=int(g['table_seed']))
hift = HvxHift(c, sd, tables=tables, exact_fp32=(sys.argv[3] == 'exact'))
gen = torch.Generator().manual_seed(int(g['mel_seed']))
mel = torch.randn(1, c.mel, int(g['T']), generator=gen)
f0_ref = torch.from_numpy(g['f0'])
# the reference's source, recomputed from ITS f0 by the oracle (2.7 M samples are not a fixture; head / tail / stride samples are)
s_ref = hift_ref.source_module(F.interpolate(f0_ref[:, None], scale_factor=float(c.upsample_total), mode='nearest').transpose(1, 2), sd, c, tables).reshape(-1)
f0 = hift.f0(mel[0]).cpu()
s = hift.source(f0_ref[0]).cpu()
wav = hift.decode(mel[0], s_ref).cpu()
wav2, _ = hift.inference(speech_feat=mel.cuda())
wav3 = hift.decode(mel[0], hift.source(f0_ref[0])).cpu()          # the reference's f0 -> OWN source -> OWN decode: two of the three stages chained
torch.save(dict(f0=f0, s=s, s_ref=s_ref, wav=wav, wav_e2e=wav2.cpu(), wav_chain=wav3), sys.argv[1])
"""


def test_hift_5632_frames_vs_reference(tmp_path):
    """cosyvoice/hifigan/generator.py:713-726 at the bench length (5632 frames -> 2 703 360 samples; the last stage is 675 841 rows of 64 channels):
    F0 predictor, source module, and the decode (production split-bf16 convolutions AND the exact-fp32 forms) fed with the reference's own source."""
    g = load_golden('hift_full.npz')
    outs = _hift_production_and_exact(_HIFT_SNIPPET, tmp_path)
    o = outs['x3']
    # F0 predictor (exact fp32 in both builds): Hz
    d_f0 = float((o['f0'] - torch.from_numpy(g['f0'][0])).abs().max())
    print('HiFT 5632 frames: f0 max |d| %.2e Hz (f0 max %.1f)' % (d_f0, g['f0'].max()))
    assert d_f0 < 1.1e-3, d_f0                                                        # measured 5.3e-4 Hz
    # the oracle's source from the reference's f0 IS the reference's source (same torch ops; other host CPU -> libm-level differences only)
    sr = o['s_ref'].numpy()
    d_or = max(np.abs(sr[::16] - g['src_s16']).max(), np.abs(sr[:32768] - g['src_head']).max(), np.abs(sr[-32768:] - g['src_tail']).max())
    assert d_or < 1e-5, d_or
    # the product's source module from the reference's f0
    s = o['s'].numpy()
    d_s = max(np.abs(s[::16] - g['src_s16']).max(), np.abs(s[:32768] - g['src_head']).max(), np.abs(s[-32768:] - g['src_tail']).max())
    print('   source module from the reference f0: max |d| %.2e (first / last 32 k and every 16th sample)' % d_s)
    assert d_s < _HIFT_BOUNDS['source'], d_s
    l2 = _check_wave('   decode(reference source), split-bf16 convolutions vs the REFERENCE', o['wav'], g, '', *_HIFT_BOUNDS['x3'])
    assert l2 < _HIFT_BOUNDS['x3_l2'], l2
    l2e = _check_wave('   decode(reference source), exact fp32 convolutions vs the REFERENCE', outs['exact']['wav'], g, '', *_HIFT_BOUNDS['exact'])
    assert l2e < _HIFT_BOUNDS['exact_l2'], l2e
    # Two stages chained: the reference's f0 -> OWN source -> OWN decode (no link of the chain is fed a reference intermediate except f0)
    l2c = _check_wave('   decode(own source(reference f0)), split-bf16 convolutions vs the REFERENCE', o['wav_chain'], g, '', *_HIFT_BOUNDS['x3'])
    assert l2c < _HIFT_BOUNDS['x3_l2'], l2c
    # End to end (own f0 -> own source -> decode).  The reference's OUTPUT is ill-conditioned in its own f0 at this length: SineGen2 accumulates rad = f0 k / 24000
    # per frame with an fp32 cumsum and multiplies the sum by 2 pi 480 (generator.py:254-260), so an f0 that differs in its LAST BIT takes another rounding path.
    # That is a measurement, not an argument: hift_full_cond.npz (make_golden_fullsize.py hift_cond) holds the distance of the reference's waveform to ITSELF with its
    # f0 moved by +-1 ulp, by +-5e-4 Hz of noise, and computed by its predictor in fp64 — sample-wise per second (1e-3 in the first second, > 1e-2 from second 5-10,
    # > 0.1 from second 43-68, up to 0.37 of a 0.99 peak) and under the phase-insensitive distances of tests/wave_metrics.py (3-7e-2: with seeded weights the network
    # is chaotic in the source phase, not merely phase-shifted).  The GPU path's own f0 differs from the reference's by 5e-4 Hz (asserted above), so its end-to-end
    # output is held to THAT band: per second to 2x the running maximum of the reference's own self-distance (+ 1e-3: the fixture's fp16 copy), and overall to 1.5x it.
    gc = load_golden('hift_full_cond.npz')
    import wave_metrics as WM
    cases = ['noise_a', 'noise_b', 'ulp_up', 'ulp_down', 'f0_fp64']
    w = o['wav_e2e'].numpy().reshape(-1)
    ref16 = g['wav_f16'].astype(np.float32)
    sec = 24000
    n = (w.size // sec) * sec
    prof = np.abs(w - ref16)[:n].reshape(-1, sec).max(axis=1)
    env = np.maximum.accumulate(np.max([gc[c + '_per_second'] for c in cases], axis=0))
    d_first = np.abs(w[:24000] - g['wav_head'][:24000]).max()
    l2 = _l2_rel(w[::16], g['wav_s16'])
    m = WM.all_metrics(w, ref16)
    worst = {k: max(float(gc[c + '_' + k]) for c in cases) for k in ('l2', 'envelope', 'stft2048_l2', 'band_energy', 'first_second', 'max')}
    over = prof / (2.0 * env + 1e-3)
    print('   end to end (own f0 and source): first second max |d| %.2e (the reference under its own f0 perturbations: up to %.2e); whole utterance max %.2e, L2 %.2e '
          '(reference: %.2e, %.2e); per second at most %.2f of the bound 2 x the reference\'s own running maximum (worst second %d)'
          % (d_first, worst['first_second'], prof.max(), l2, worst['max'], worst['l2'], over.max(), int(over.argmax())))
    print('   phase-insensitive distances, GPU vs reference / the reference vs itself (worst of 5 perturbations): frame-RMS envelope %.2e / %.2e, |STFT 2048| L2 %.2e / %.2e, '
          'band energies %.2e / %.2e' % (m['envelope'], worst['envelope'], m['stft2048_l2'], worst['stft2048_l2'], m['band_energy'], worst['band_energy']))
    assert (prof <= 2.0 * env + 1e-3).all(), (int(over.argmax()), float(over.max()))
    assert d_first < 2.0 * worst['first_second'] and l2 < 1.5 * worst['l2'], (d_first, l2)
    assert m['envelope'] < 1.5 * worst['envelope'] and m['stft2048_l2'] < 1.5 * worst['stft2048_l2'] and m['band_energy'] < 1.5 * worst['band_energy'], m


# measured on MI355X (round 5), bounds <= 2x: see DESIGN.md §3.  CONTRACT 1e-3 for the vocoder stages: every stage-wise number below meets it (asserted far inside);
# the end-to-end waveform is held to the reference's own conditioning band instead (hift_full_cond.npz), which is wider than 1e-3 for the reference itself.
#   f0 5.3e-4 Hz; source (reference f0) 3.0e-8; decode(reference source): split-bf16 2.1e-4 max / 1.5e-4 L2 (fp16 copy 3.9e-4), exact fp32 4.6e-5 / 2.8e-5 (2.7e-4)
_HIFT_BOUNDS = {'source': 1e-6, 'x3': (4.5e-4, 8e-4), 'x3_l2': 3e-4, 'exact': (1e-4, 6e-4), 'exact_l2': 6e-5}


# ---- LM on a 3300-row prefix -------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope='module')
def llm_full():
    from flowmirror_hydravox_amd import cv3_config
    from flowmirror_hydravox_amd import weights as W
    g = load_golden('llm_full.npz')
    c = cv3_config().llm
    sd = W.make_llm_state(c, seed=int(g['weight_seed']), init='fan_in', with_lm_head=True)
    assert state_checksum(sd) == str(g['weight_sha'])
    top_p, top_k, win, tau = g['sampling']
    return g, c, sd, dict(top_p=float(top_p), top_k=int(top_k), win_size=int(win), tau_r=float(tau))


def _mk_llm(c, sd, sampling, dtype, max_batch, max_ctx):
    from functools import partial
    from flowmirror_hydravox_amd.llm import HvxLLM
    from flowmirror_hydravox_amd.sampling import ras_sampling
    return HvxLLM(c, sd, dtype=dtype, max_batch=max_batch, max_ctx=max_ctx, sampling=partial(ras_sampling, **sampling))


def _decode_last_row(llm, c, enc, S, K):
    """prefill enc[:-1] into S slots, then ONE decode step of enc[-1] in every slot (the wide-grid kernels at S >= 17 rows): log-probs (S, K, V)"""
    dev = llm.device
    n = len(enc) - 1
    llm._bind(S, max(n, S * K))
    tok = torch.tensor(enc[:-1], dtype=torch.int32, device=dev)
    scratch = torch.empty(1, K, c.vocab, dtype=torch.float32, device=dev)
    for s in range(S):
        llm._forward(1, n, tok, torch.tensor([s, 0, n, n, n - 1], dtype=torch.int32, device=dev), K, scratch)
    step = torch.full((S,), int(enc[-1]), dtype=torch.int32, device=dev)
    ctrl = torch.tensor([list(range(S)), [n] * S, [1] * S, [n + 1] * S, list(range(S))], dtype=torch.int32).reshape(-1).to(dev)
    logp = torch.empty(S, K, c.vocab, dtype=torch.float32, device=dev)
    llm._forward(S, 1, step, ctrl, K, logp)
    torch.cuda.synchronize()
    return logp.cpu()


def test_llm_3300_row_prefix_fp32_vs_reference(llm_full):
    """llm_multi_head_v3.py:248-260, 886-888 at the END of a 512-char utterance's context range: first-step hidden / log-probs of all 5 heads after a
    3300-row prefill, the same log-probs from ONE decode step over a 3299-row cache (7 attention splits of 512 keys / 13 of 256), and the reference's
    own uncached K = 2 generation from that context, ids bit-exact."""
    g, c, sd, sampling = llm_full
    llm = _mk_llm(c, sd, sampling, torch.float32, max_batch=2, max_ctx=3392)
    text, ps = torch.from_numpy(g['text']), torch.from_numpy(g['pspeech'])
    enc = llm._encode_prefix(text, None, ps)
    assert len(enc) == 3300
    llm.inference_head_num = 5
    logp, y = llm.prefill_logp(enc)
    e = (_scale_rel(y.cpu().numpy(), g['y_last']), float(np.abs(logp.cpu().numpy() - g['logps']).max()))
    print('24 layers fp32, 3300-row prefix vs the REFERENCE: hidden %.1e of its scale, log-probs %.1e abs' % e)
    assert e[0] < 1e-5 and e[1] < 1e-4, e                                             # measured 1.9e-6 / 1.7e-5
    dec = _decode_last_row(llm, c, enc, 1, 5)[0].numpy()
    e2 = float(np.abs(dec - g['logps']).max())
    print('   one decode step over the 3299-row cache: log-probs %.1e abs' % e2)
    assert e2 < 1e-4, e2                                                              # measured 1.3e-5
    llm.inference_head_num = int(g['K'])
    want = g['tokens'].tolist()
    got = []
    for tok in llm.inference(text=text[None], text_len=torch.tensor([text.numel()], dtype=torch.int32), prompt_text=torch.zeros(1, 0, dtype=torch.int32),
                             prompt_text_len=torch.tensor([0], dtype=torch.int32), prompt_speech_token=ps[None], prompt_speech_token_len=torch.tensor([ps.numel()], dtype=torch.int32),
                             embedding=torch.zeros(0, 192), max_token_text_ratio=0.15, min_token_text_ratio=2, seed=int(g['seed'])):       # (max_len 76 fits max_ctx; min_len 1024 as minted: no EOS may be drawn)
        got.append(int(tok))
        if len(got) >= len(want):
            break
    assert got == want, (got, want)


def test_llm_fp32_64_slot_grid_ids_vs_reference_beside_63_live_sequences(llm_full):
    """The geometry of bench.py's `ids_exact_mode` (fp32 LM, the continuous engine, 64 slots x 2 heads = 128 rows per step: gemm_dec.hip's A-stationary form on the
    exact fp32 MFMAs, fp32 KV cache, 512-key attention splits), where the claim "speech-token ids bit-exact against the reference" is printed: the request of
    llm_full.npz (3300-row prefix, the reference's own uncached K = 2 generation, llm_multi_head_v3.py:871-922) joins the grid as its 64th sequence, beside 63 live
    sequences at contexts > 1024 (1042-row prefixes: eight of them per grouped prefill forward) that keep decoding for as long as it does."""
    g, c, sd, sampling = llm_full
    llm = _mk_llm(c, sd, sampling, torch.float32, max_batch=64, max_ctx=3392)
    llm.inference_head_num = int(g['K'])
    assert llm.inference_head_num == 2
    text, ps = torch.from_numpy(g['text']), torch.from_numpy(g['pspeech'])
    gen = torch.Generator().manual_seed(4100)
    fillers = [dict(text=torch.randint(0, c.text_vocab, (1040,), generator=gen, dtype=torch.int32), seed=7000 + i, tag=('f', i),
                    max_token_text_ratio=0.08, min_token_text_ratio=0.08) for i in range(63)]               # 83 tokens each: alive for all of the 38 steps of the pinned request
    main = dict(text=text, prompt_text=torch.zeros(0, dtype=torch.int32), prompt_speech_token=ps, seed=int(g['seed']), tag='ref',
                max_token_text_ratio=0.15, min_token_text_ratio=2)                                           # (max_len 76 fits max_ctx; min_len 1024 as minted: no EOS may be drawn)
    got = dict(llm.generate_stream(iter(fillers + [main]), n_slots=64))
    st = llm.last_stats
    want = g['tokens'].tolist()
    print('fp32 LM, 64-slot continuous grid, K = 2: %d requests, %d steps, mean live sequences %.1f, mean context %.0f rows; the pinned request emitted %d ids, the first %d == the REFERENCE\'s'
          % (st['requests'], st['steps'], st['mean_active_sequences'], st['mean_ctx'], len(got['ref']), len(want)))
    assert st['requests'] == 64 and all(len(got[('f', i)]) == 83 for i in range(63))
    assert len(got['ref']) == 76 and got['ref'][:len(want)] == want, (got['ref'][:len(want)], want)
    assert st['mean_active_sequences'] > 48 and st['mean_ctx'] > 1024, st
    # ... and alone in the same 64-slot grid: the ids of a request do not depend on what shares the grid with it (all 76)
    alone = dict(llm.generate_stream(iter([dict(main)]), n_slots=64))
    assert alone['ref'] == got['ref']


def test_llm_fp32_64_slot_grid_is_batch_invariant_over_a_whole_512_char_stream(llm_full):
    """One 512-char utterance of the bench (512 text ids -> 2816 speech tokens, head_num 2: 1408 decode steps, contexts 514 -> 3330, crossing every 512-key split
    boundary of the decode attention) through the fp32 64-slot grid ALONE and beside 63 other 512-char utterances (the `ids_exact_mode` job itself): the same 2816 ids.
    Together with the test above (ids == the reference's inside the full grid) this carries the reference pin to every row of that job."""
    g, c, sd, sampling = llm_full
    llm = _mk_llm(c, sd, sampling, torch.float32, max_batch=64, max_ctx=2 + 512 + 2816 + 2 + 32)
    llm.inference_head_num = 2

    def req(i):
        return dict(text=torch.randint(0, c.text_vocab, (512,), generator=torch.Generator().manual_seed(4200 + i), dtype=torch.int32), seed=9000 + i, tag=i,
                    max_token_text_ratio=5.5, min_token_text_ratio=5.5)
    alone = dict(llm.generate_stream(iter([req(0)]), n_slots=64))
    assert len(alone[0]) == 2816
    steps_alone = llm.last_stats['steps']
    crowd = dict(llm.generate_stream(iter([req(i) for i in range(64)]), n_slots=64))
    st = llm.last_stats
    print('fp32 LM, one 512-char stream alone (%d steps) vs inside the full 64-slot grid (%d steps, mean live sequences %.1f, mean context %.0f): %d ids equal'
          % (steps_alone, st['steps'], st['mean_active_sequences'], st['mean_ctx'], len(alone[0])))
    assert st['steps'] >= 1024 and st['mean_active_sequences'] > 60
    assert all(len(crowd[i]) == 2816 for i in range(64))
    assert crowd[0] == alone[0]
    assert len({tuple(v) for v in crowd.values()}) == 64                                                   # (64 different streams: nobody decoded somebody else's rows)


def test_llm_3300_row_context_bf16_wide_grid_vs_reference(llm_full):
    """The production decode path at that context: 40 slots x 2 heads = 80 rows (gemm_dec.hip over fragment-order activations, the fragment-order KV cache,
    512-key attention splits x 7) — one decode step over 3299 cached rows in every slot against the REFERENCE's fp32 log-probs; and the bf16 prefill forms."""
    g, c, sd, sampling = llm_full
    llm = _mk_llm(c, sd, sampling, torch.bfloat16, max_batch=40, max_ctx=3392)
    text, ps = torch.from_numpy(g['text']), torch.from_numpy(g['pspeech'])
    enc = llm._encode_prefix(text, None, ps)
    llm.inference_head_num = 5
    logp, y = llm.prefill_logp(enc)
    top = torch.from_numpy(g['logps']).topk(25, dim=-1).indices                   # the tokens the sampler can pick (top_k <= 25)
    ref_top = torch.from_numpy(g['logps']).gather(-1, top)
    e = (_scale_rel(y.cpu().numpy(), g['y_last']), float((logp.cpu().gather(-1, top) - ref_top).abs().max()), float(np.abs(logp.cpu().numpy() - g['logps']).max()))
    print('24 layers bf16, 3300-row prefill vs the REFERENCE (fp32): hidden %.2e of its scale, log-probs of the top-25 %.3f abs, all %.3f abs' % e)
    assert e[0] < _LLM_BOUNDS['hidden'] and e[1] < _LLM_BOUNDS['top'] and e[2] < _LLM_BOUNDS['all'], e
    K, S = 2, 40
    dec = _decode_last_row(llm, c, enc, S, K)
    assert torch.isfinite(dec).all()
    for s in range(1, S):                                                             # every slot holds the same cache: rows never mix
        assert torch.equal(dec[s], dec[0]), s
    d_top = float((dec[0].gather(-1, top[:K]) - ref_top[:K]).abs().max())
    d_all = float((dec[0] - torch.from_numpy(g['logps'][:K])).abs().max())
    flips = int((dec[0].argmax(-1) != torch.from_numpy(g['logps'][:K]).argmax(-1)).sum())
    print('   80-row decode step over 3299 cached rows (7 splits of 512 keys): top-25 log-probs %.3f abs, all %.3f abs, argmax flips %d of %d' % (d_top, d_all, flips, K))
    assert d_top < _LLM_BOUNDS['top'] and d_all < _LLM_BOUNDS['all'], (d_top, d_all)


# measured on MI355X (round 5), bounds <= 2x: see DESIGN.md §3
#   measured: prefill hidden 8.9e-3, top-25 log-probs 0.052, all 0.079; 80-row decode step 0.046 / 0.064, no argmax flip
_LLM_BOUNDS = {'hidden': 1.8e-2, 'top': 0.1, 'all': 0.16}
