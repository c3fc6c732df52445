"""Full-size properties at the shapes of BASELINE.json configs[1] (HydraVox-CV3 widths, 5632 mel frames per utterance, 4 utterances x CFG 2
per estimator call = the launches of `python bench.py`).  The oracle cannot run these sizes in seconds, so the production bf16 kernels
(256-tile GEMM with its compile-time epilogue modes, 256-row LDS-staged attention with the pre-scaled fixed-reference softmax, two-rows-per-wave
LayerNorm) are held against the exact-fp32 forms of the same library (fp32 MFMA tiles, generic attention) — different kernels, which the
small-size tests pin to the oracle and to the reference-minted fixtures (tests/test_gpu_models.py, tests/test_gpu_cv3w.py)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a - b).norm() / b.norm())


def test_dit_estimator_at_the_bench_shape_bf16_forms_vs_fp32_forms():
    from flowmirror_hydravox_amd import cv3_config
    from flowmirror_hydravox_amd import weights as W
    from flowmirror_hydravox_amd.flow import HvxFlow
    cfg = cv3_config().flow
    sd = W.make_flow_state(cfg, seed=1987, init='fan_in')
    T, B = 5632, 8
    gen = torch.Generator().manual_seed(3)
    x, mu, cond = (torch.randn(B, cfg.mel, T, generator=gen) for _ in range(3))
    spk = torch.randn(B, cfg.mel, generator=gen)
    t = torch.full((B,), 0.35)
    lens = [T, 5000, T, T, T, 4100, T, T]                     # two padded entries: key-padding masks at full size
    mask = torch.zeros(B, 1, T)
    for i, n in enumerate(lens):
        mask[i, :, :n] = 1
    outs = {}
    for dt in (torch.float32, torch.bfloat16):
        flow = HvxFlow(cfg, sd, dtype=dt, max_t=T + 64)
        outs[dt] = flow.estimator(x, mask, mu, t, spk, cond).cpu()
        del flow
        torch.cuda.empty_cache()
    assert torch.isfinite(outs[torch.float32]).all() and torch.isfinite(outs[torch.bfloat16]).all()
    rels = [_rel(outs[torch.bfloat16][i, :, :n], outs[torch.float32][i, :, :n]) for i, n in enumerate(lens)]
    print('bf16 vs fp32 forms, 22 blocks, T = %d: relative error per batch entry %s' % (T, ['%.2e' % r for r in rels]))
    # measured 1.7e-3 per entry in the default (reference-precision) bf16 mode — 4.8e-3..5.6e-3 with plain bf16 operands everywhere; bound = 2x measured.
    # (The same shape against the REFERENCE itself: tests/test_gpu_refpin.py.)
    assert max(rels) < 4e-3, rels
    # rows of the full-length entries do not depend on their neighbours in the batch: entry 0 alone (704 GEMM tiles -> 176, attention grid / 4)
    flow = HvxFlow(cfg, sd, dtype=torch.bfloat16, max_t=T + 64)
    alone = flow.estimator(x[:2], mask[:2], mu[:2], t[:2], spk[:2], cond[:2]).cpu()
    assert _rel(alone[0], outs[torch.bfloat16][0]) < 1e-6 and _rel(alone[1, :, :5000], outs[torch.bfloat16][1, :, :5000]) < 1e-6


def test_cfm_solve_of_four_512_char_utterances_bf16_vs_fp32_forms():
    """One acoustic batch of the bench: 4 token streams (2816, 2816, 2500, 2050 tokens -> 5632 ... 4100 frames) through the padded 10-step
    CFG solve (hvx_cfm_solve_batch), production bf16 forms against the exact-fp32 forms of the same library; the mel of the north star
    ('within 1e-3 rel bf16' is stated for the reference's own CPU path; the residual here is the bf16 operand rounding of 220 estimator
    blocks per utterance, printed)."""
    from flowmirror_hydravox_amd import cv3_config
    from flowmirror_hydravox_amd import weights as W
    from flowmirror_hydravox_amd.flow import HvxFlow
    cfg = cv3_config().flow
    sd = W.make_flow_state(cfg, seed=1987, init='fan_in')
    g = torch.Generator().manual_seed(5)
    lens = [2816, 2816, 2500, 2050]
    toks = [torch.randint(0, cfg.vocab, (n,), generator=g, dtype=torch.int32).cuda() for n in lens]
    embs = [torch.randn(cfg.spk_embed_dim, generator=g).cuda() for _ in lens]
    mels = {}
    for dt in (torch.float32, torch.bfloat16):
        flow = HvxFlow(cfg, sd, dtype=dt, max_t=2 * max(lens) + 64)
        mels[dt] = [m.cpu() for m in flow.inference_batch(toks, embs)]
        del flow
        torch.cuda.empty_cache()
    rels = []
    for n, a, b in zip(lens, mels[torch.bfloat16], mels[torch.float32]):
        assert tuple(a.shape) == (1, cfg.mel, 2 * n) == tuple(b.shape) and torch.isfinite(a).all() and torch.isfinite(b).all()
        rels.append(_rel(a, b))
    print('10-step CFG solve, bf16 vs fp32 forms: relative mel error per utterance %s' % ['%.2e' % r for r in rels])
    # measured 1.4e-3 per utterance in the default mode (3.2e-3 with plain bf16 operands); bound = 2x measured.  Against the REFERENCE: tests/test_gpu_refpin.py.
    assert max(rels) < 3e-3, rels


_HIFT_SNIPPET = r"""
import sys, torch
sys.path.insert(0, sys.argv[2])
from flowmirror_hydravox_amd import cv3_config
from flowmirror_hydravox_amd import weights as W
from flowmirror_hydravox_amd.hift import HvxHift
from flowmirror_hydravox_amd import _lib
c = cv3_config().hift
mode = sys.argv[3]
if mode == 'x3_tiled':
    _lib.set_option('conv64_resident', 0)
hift = HvxHift(c, W.make_hift_state(c, seed=1988, init='fan_in'), exact_fp32=(mode == 'exact'))
g = torch.Generator().manual_seed(11)
mel = (torch.randn(1, c.mel, 5632, generator=g) * 1.5 - 4.0).cuda()
wav, _ = hift.inference(speech_feat=mel)
torch.save(wav.cpu(), sys.argv[1])
"""


def test_hift_of_a_512_char_utterance_split_bf16_convs_vs_exact_fp32_convs(tmp_path):
    """The vocoder at the bench length (5632 mel frames -> 112.6 s of audio, HiFT base 512): the production convolutions (fp32 operands as
    (hi, lo) bf16 pairs, gemm_x3.hip, with the ResBlock epilogue modes) against the exact fp32-MFMA convolutions of the same library
    (HvxHift(exact_fp32=True) = hvx_hift_config.exact_fp32: the form the small-size tests pin to the reference), and the tiled form of the 64-channel convolutions
    (option conv64_resident = 0).  One process per form: each holds ~6 GB of workspace."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    wavs = []
    for name in ('x3', 'exact', 'x3_tiled'):
        out = str(tmp_path / (name + '.pt'))
        r = subprocess.run([sys.executable, '-c', _HIFT_SNIPPET, out, root, name], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:]
        wavs.append(torch.load(out))
    a, b, c = wavs
    # the last stage's 64-channel convolutions: rows resident in LDS across the taps (conv64_x3p_kernel) vs the tiled plane-pair form — the same
    # products summed in the same order
    assert torch.equal(a, c), float((a - c).abs().max())
    assert a.shape == b.shape and a.numel() == 5632 * 480 and torch.isfinite(a).all() and torch.isfinite(b).all()
    rel = _rel(a, b)
    print('HiFT 5632 frames: split-bf16 vs exact fp32 convolutions, relative waveform difference %.2e, max |diff| %.2e' % (rel, float((a - b).abs().max())))
    # measured 2.2e-4 (the harmonic source integrates a phase over 2.7 M samples; the small-size bound per convolution is 2e-5)
    assert rel < 1e-3, rel


@pytest.mark.parametrize('S', [32, 64])
def test_llm_decode_grid_at_full_depth_vs_teacher_forced_prefill(S):
    """The 24-layer CV3 LM, S sequences x 2 heads = the 64- / 128-row decode grid forms of the bench (skinny GEMMs in 64-row chunks, mid-M
    split-K, GQA-packed split attention + combine) at contexts 1150..1970: the log-probs of ONE decode step over KV caches filled by prefill must be the
    log-probs of a teacher-forced prefill of prefix + the step's tokens —
      * bf16 decode step vs bf16 prefill: the same arithmetic through different kernels (tiled prefill forms) -> bf16 rounding-order noise;
      * bf16 decode step vs the exact-fp32 prefill forms (what the small-size tests pin to the oracle / the reference): the bf16 contract."""
    from flowmirror_hydravox_amd import cv3_config
    from flowmirror_hydravox_amd import weights as W
    from flowmirror_hydravox_amd.llm import HvxLLM
    c = cv3_config().llm
    sd = W.make_llm_state(c, seed=1986, init='fan_in')
    K = 2
    g = torch.Generator().manual_seed(9)
    prefixes, steps = [], []
    for s in range(S):
        n_text, n_sp = 150 + 4 * s, 1000 + 9 * s                                    # contexts 1152 .. 1971 rows
        text = torch.randint(0, c.text_vocab, (n_text,), generator=g, dtype=torch.int32)
        sp = torch.randint(0, c.speech_tokens, (n_sp + K,), generator=g, dtype=torch.int32)
        prefixes.append((text, sp[:n_sp]))
        steps.append(sp[n_sp:])
    probe = [0, 7, 19, S - 1]                                                         # sequences teacher-forced for the comparison
    ref = {}
    for dt in (torch.float32, torch.bfloat16):
        llm = HvxLLM(c, sd, dtype=dt, max_batch=S, max_ctx=2048, inference_head_num=K)
        for s in probe:
            text, sp = prefixes[s]
            logp, y = llm.prefill_logp(llm._encode_prefix(text, None, torch.cat([sp, steps[s]])), head_k=K)
            ref[(dt, s)] = (logp.cpu(), y.cpu())
        if dt == torch.bfloat16:
            dev = llm.device
            llm._bind(S, 2048)
            lens = []
            for s in range(S):                                                       # prefill every prefix into its own KV slot
                text, sp = prefixes[s]
                enc = llm._encode_prefix(text, None, sp)
                n = len(enc)
                lens.append(n)
                tok = torch.tensor(enc, dtype=torch.int32, device=dev)
                ctrl = torch.tensor([s, 0, n, n, n - 1], dtype=torch.int32, device=dev)
                llm._forward(1, n, tok, ctrl, K, torch.empty(1, K, c.vocab, dtype=torch.float32, device=dev))
            tok = torch.cat(steps).to(dev)
            ctrl = torch.tensor([list(range(S)), lens, [K] * S, [n + K for n in lens], [i * K + K - 1 for i in range(S)]], dtype=torch.int32).reshape(-1).to(dev)
            logp = torch.empty(S, K, c.vocab, dtype=torch.float32, device=dev)
            llm._forward(S, K, tok, ctrl, K, logp)
            torch.cuda.synchronize()
            dec = logp.cpu()
        del llm
        torch.cuda.empty_cache()
    worst = [0.0, 0.0, 0.0]
    for s in probe:
        lb, lf = ref[(torch.bfloat16, s)][0], ref[(torch.float32, s)][0]
        top = lf.topk(25, dim=-1).indices                                            # the tokens the sampler can pick (top_k 25 of the reference)
        d_same = (dec[s].gather(-1, top) - lb.gather(-1, top)).abs().max().item()
        d_fp32 = (dec[s].gather(-1, top) - lf.gather(-1, top)).abs().max().item()
        agree = (dec[s].argmax(-1) == lf.argmax(-1)).all().item()
        worst = [max(worst[0], d_same), max(worst[1], d_fp32), worst[2] + (0 if agree else 1)]
    print('24 layers, %d-row decode grid: |dlogp| over the top-25 tokens vs the bf16 prefill forms %.3f, vs the fp32 forms %.3f; argmax flips %d of %d'
          % (S * K, worst[0], worst[1], worst[2], len(probe)))
    assert torch.isfinite(dec).all()
    # measured 0.034 / 0.064 at 64 rows (log-probs of the 25 most likely of 6761 tokens, 24 layers + the 22016-wide MTP heads in bf16)
    assert worst[0] < 0.1 and worst[1] < 0.2, worst


@pytest.mark.parametrize('S,K', [(20, 2), (64, 2), (25, 4), (100, 2), (40, 4), (50, 1)])
def test_wide_grid_decode_gemm_form_agrees_with_the_generic_kernels(S, K):
    """gemm_dec.hip (A-stationary / weight-ring GEMMs over fragment-order activations, 33..256 rows) against the generic skinny kernels on
    row-major activations (option dec_gemm = 0): one decode step of a 2-layer CV3-width LM over a random KV cache, ragged positions and row counts
    (40 rows: a partial last row tile; 128: two 64-row chunks; 100 and 200: partial chunks, three and four chunks).  The MTP heads take the form
    too from 33 sequences on — gate / up with every head's rows as its own fragment-order matrix (40 sequences: padded to 48 rows per head), the
    shared output projection over the stacked rows of all heads (from 33 ROWS on: 20 sequences x 2 heads already).  The two differ in the fp32
    summation order of every GEMM (K is not split inside a workgroup any more) and in where bf16 roundings fall after it: log-probs of the
    sampler's candidates within 3e-2 (measured 1.2e-2), appended K / V rows within 2e-2."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    import dec_ab
    r = dec_ab.compare(seqs=S, heads=K, ctx=300, layers=2)
    print(S, K, r)
    assert r['finite'] and r['max_logp'] < 3e-2 and r['mean_logp'] < 5e-3 and r['kv_max'] < 2e-2 and r['argmax'] > 0.95, r


@pytest.mark.parametrize('S,K,ctx', [(1, 1, 700), (1, 2, 1536), (3, 4, 700), (8, 2, 1100), (3, 5, 600)])
def test_narrow_grid_o_proj_from_attention_partials_is_bit_identical_to_the_combine_launch(S, K, ctx):
    """Decode grids of <= 16 rows (a single request, the literal batch of 8 x 2 heads): the o_proj builds its activation fragments from the attention's
    key-split partials (gemm_skinny.hip: combined_pair; option dec_fuse_rows) instead of reading what attn_combine_kernel wrote — the same operations in the
    same order, so one decode step of a 2-layer CV3-width LM over a random KV cache (ragged positions and row counts, 3..6 live splits) must give the same
    BITS in both forms, bf16 and fp32.  (3 x 5: 35 GQA rows per KV head — no key split, nothing fused: the option must then change nothing.)"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    import fuse_ab
    for dt in (torch.bfloat16, torch.float32):
        r = fuse_ab.one(S, K, ctx, 2, dt, steps=0, ragged=True)
        print(r)
        assert r['finite'] and r['same_logp'] and r['same_kv'], r
