"""HvxModelManager + the top-level synthesis functions — the drop-in surface consumed by the reference's worker.

Mirrors server/model_utils/infer_speech_model.py:
  * ModelManager.load_models(args)  (:50-143)  args.{config, model_dir, bf16, fp16, cpu}; loads `llm.pt`, `flow.pt`, `hift.pt`
    (flat state dicts, keys epoch/step/_original_metadata/_conversion_info dropped :80-89, strict load :92-94);
    sets .models = {'llm','flow','hift'}, .configs['sample_rate'], .frontend, .device, .is_loaded, .zero_shot_speakers
  * ModelManager.load_pt(llm_pt, flow_pt) -> {"status": "success"|"error", "message": str}, never raises  (:169-184)
  * inference_zero_shot / inference_tts / text_to_speech  (:523-606, :612-690, :743-820)
Model hyper-parameters: the reference builds its modules from `<model_dir>/hydravox.yaml` through HyperPyYAML (:59-62), which is
not part of this build; yaml_config.py reads the plain numbers of that file (constructor arguments of the `!new:` nodes, `!ref`s
resolved, Qwen2 sizes from `CosyVoice-BlankEN/config.json`).  `<model_dir>/hvx_config.json`, when present, takes precedence;
with neither file the [ASSUMED-CV3] preset applies.  The text / audio frontend (ONNX tokenizers, text normalisation) is out of scope: `frontend` is any
object with the reference's `text_normalize`, `frontend_sft`, `frontend_zero_shot` methods (e.g. the reference's own
CosyVoiceFrontEnd), injected by the caller.
"""
import json
import logging
import os
import time

import torch

from .config import HvxConfig, LLMConfig, FlowConfig, HiftConfig, cv3_config
from .flow import HvxFlow
from .hift import HvxHift
from .llm import HvxLLM
from .ops import resample_linear
from .weights import DROP_KEYS

logger = logging.getLogger('hvx')


def _load_config(model_dir):
    """-> (HvxConfig, extras): extras = the yaml's non-dimension settings (sampling defaults, inference_head_num), {} otherwise"""
    p = os.path.join(model_dir or '', 'hvx_config.json')
    if model_dir and os.path.exists(p):
        with open(p) as f:
            d = json.load(f)
        return HvxConfig(llm=LLMConfig(**d.get('llm', {})), flow=FlowConfig(**d.get('flow', {})), hift=HiftConfig(**d.get('hift', {})),
                         sample_rate=d.get('sample_rate', 24000)), {}
    if model_dir:
        from .yaml_config import config_from_model_dir
        got = config_from_model_dir(model_dir)                 # <model_dir>/hydravox.yaml, as the reference (:59-62)
        if got is not None:
            return got
    return cv3_config(), {}


def _load_pt(path):
    sd = torch.load(path, map_location='cpu')
    for k in DROP_KEYS:
        if k in sd:
            sd.pop(k)
    return sd


class HvxModelManager:
    def __init__(self, frontend_factory=None):
        self.models = None
        self.frontend = None
        self.configs = None
        self.device = None
        self.is_loaded = False
        self.zero_shot_speakers = None
        self.hvx_config = None
        self._frontend_factory = frontend_factory

    def load_models(self, args):
        if self.is_loaded:                                   # idempotent (:52-54)
            logger.info('models already loaded, skipping')
            return
        if getattr(args, 'cpu', False) or not torch.cuda.is_available():
            raise ValueError('HvxModelManager runs on MI355X only: there is no CPU path (TTS_CPU is not supported)')
        cfg, extras = _load_config(args.model_dir)
        self.hvx_config = cfg
        self.device = 'cuda'
        # precision policy of the reference (:101-118): llm bf16 / flow half / hift fp32.  fp16 requests run in bf16 as well:
        # libhvx computes in bf16 (or fp32), with fp32 accumulation and an fp32 residual stream.
        # optional fp8 head weights (llm.HvxLLM: head_mlp_fp8): args.head_fp8 or $HVX_HEAD_FP8=1
        head_fp8 = bool(getattr(args, 'head_fp8', False)) or os.environ.get('HVX_HEAD_FP8', '').strip().lower() in ('1', 'true', 'yes', 'on')
        llm = HvxLLM(cfg.llm, None, dtype=torch.bfloat16, head_mlp_fp8=head_fp8)
        if extras.get('sampling'):                           # the yaml's `sampling: !name:...ras_sampling` keyword defaults
            from functools import partial
            from .sampling import ras_sampling
            llm.sampling = partial(ras_sampling, **extras['sampling'])
        if extras.get('inference_head_num'):
            llm.inference_head_num = int(extras['inference_head_num'])
        flow = HvxFlow(cfg.flow, None, dtype=torch.bfloat16)
        hift = HvxHift(cfg.hift, None)
        # optional packed-weight cache: args.packed_cache or $HVX_PACKED_CACHE names a directory that holds the
        # device-ready tensors of each `.pt` (checkpoint.py); without it the `.pt` files are read and packed on every start
        self.packed_cache = getattr(args, 'packed_cache', None) or os.environ.get('HVX_PACKED_CACHE') or None
        self.load_report = {}
        for name, model in (('llm', llm), ('flow', flow), ('hift', hift)):
            self.load_report[name] = self._load_into(model, os.path.join(args.model_dir, name + '.pt'))
        llm.bf16, flow.bf16, llm.fp16, flow.fp16 = True, True, False, False
        self.models = {'llm': llm, 'flow': flow, 'hift': hift}
        self.configs = {'sample_rate': cfg.sample_rate}
        if self._frontend_factory is not None:
            self.frontend = self._frontend_factory(args, cfg)
        spk = os.path.join(args.model_dir, 'zero_shot_speakers_16k.pt')
        self.zero_shot_speakers = torch.load(spk) if os.path.exists(spk) else None
        self.is_loaded = True
        logger.info('models loaded')

    def _load_into(self, model, pt_path):
        if getattr(self, 'packed_cache', None):
            from .checkpoint import load_or_pack
            return load_or_pack(model, pt_path, self.packed_cache, _load_pt)
        model.load_state_dict(_load_pt(pt_path))
        return 'packed'

    def load_pt(self, llm_pt, flow_pt):
        try:
            self._load_into(self.models['llm'], llm_pt)
            self.models['llm'].bf16, self.models['llm'].fp16 = True, False
            self._load_into(self.models['flow'], flow_pt)
            self.models['flow'].bf16, self.models['flow'].fp16 = True, False
            return {'status': 'success', 'message': 'model weights loaded'}
        except Exception as e:                               # never raises (:181-184)
            logger.error('load_pt failed: %s', e)
            return {'status': 'error', 'message': str(e)}

    def get_available_speakers(self):
        if not self.frontend or not hasattr(self.frontend, 'spk2info'):
            return []
        return list(self.frontend.spk2info.keys())


def _synthesize(model_manager, model_input, speed, zero_shot):
    """llm -> flow -> hift for one utterance (the body shared by :548-606 and :630-690)."""
    dev = model_manager.device
    start = time.time()
    kw = dict(text=model_input['text'], text_len=model_input['text_len'], embedding=model_input['llm_embedding'])
    if zero_shot:
        kw.update(prompt_text=model_input['prompt_text'], prompt_text_len=model_input['prompt_text_len'],
                  prompt_speech_token=model_input['llm_prompt_speech_token'],
                  prompt_speech_token_len=model_input['llm_prompt_speech_token_len'])
    else:
        kw.update(prompt_text=torch.tensor([], dtype=torch.int32), prompt_text_len=torch.tensor([0], dtype=torch.int32),
                  prompt_speech_token=None, prompt_speech_token_len=torch.tensor([0], dtype=torch.int32))
    tokens = [t for t in model_manager.models['llm'].inference(**kw)]
    llm_time = time.time() - start
    tps = len(tokens) / llm_time if llm_time > 0 else 0
    token_tensor = torch.tensor(tokens).unsqueeze(0).to(dev)
    fkw = dict(token=token_tensor, token_len=torch.tensor([token_tensor.shape[1]], dtype=torch.int32), streaming=False, finalize=True)
    if zero_shot:
        fkw.update(prompt_token=model_input['flow_prompt_speech_token'], prompt_token_len=model_input['flow_prompt_speech_token_len'],
                   prompt_feat=model_input['prompt_speech_feat'], prompt_feat_len=model_input['prompt_speech_feat_len'],
                   embedding=model_input['flow_embedding'])
    else:
        fkw.update(embedding=model_input['flow_embedding'].unsqueeze(0))
    tts_mel, _ = model_manager.models['flow'].inference(**fkw)
    if speed <= 0:
        raise ValueError('Invalid speed: %s' % speed)
    if speed != 1.0:
        tts_mel = resample_linear(tts_mel, max(1, int(tts_mel.shape[2] / speed)))
    tts_speech, _ = model_manager.models['hift'].inference(speech_feat=tts_mel)
    total = time.time() - start
    audio_len = tts_speech.shape[-1] / 24000
    logger.info('inference done, total %.2fs, TPS %.2f, RTF %.4f', total, tps, total / audio_len if audio_len else 0.0)
    return tts_speech.cpu()


def synthesize_many(model_manager, model_inputs, zero_shot, speeds=None, seeds=None):
    """Batched form of `_synthesize`: the frontend outputs of several requests are decoded by the LM in lock-step
    (one weight stream serves every request of the batch, which is where the decode roofline is), then each goes through flow and the
    vocoder.  `zero_shot[i]` says whether request i carries a prompt.  Every request samples from its OWN generator: `seeds[i]`, or a
    seed drawn here from torch's global generator (with one global generator the draws of concurrent requests would interleave; the
    per-request result equals `_synthesize` run alone with that seed).  -> list of cpu waveforms (1, L)"""
    n = len(model_inputs)
    speeds = [1.0] * n if speeds is None else list(speeds)
    if any(s <= 0 for s in speeds):
        raise ValueError('Invalid speed: %s' % min(speeds))
    if seeds is None:
        seeds = [int(v) for v in torch.randint(0, 2 ** 31 - 1, (n,)).tolist()]
    llm, flow, hift = (model_manager.models[k] for k in ('llm', 'flow', 'hift'))
    dev = model_manager.device
    texts = [mi['text'].reshape(-1) for mi in model_inputs]
    ptexts = [mi['prompt_text'].reshape(-1) if z else None for mi, z in zip(model_inputs, zero_shot)]
    pspeech = [mi['llm_prompt_speech_token'].reshape(-1) if z else None for mi, z in zip(model_inputs, zero_shot)]
    start = time.time()
    toks = llm.generate_batch(texts, prompt_texts=ptexts if any(zero_shot) else None, prompt_speech_tokens=pspeech if any(zero_shot) else None,
                              seeds=seeds)
    llm_time = time.time() - start
    outs = []
    for mi, z, t, speed in zip(model_inputs, zero_shot, toks, speeds):
        token = torch.tensor(t).unsqueeze(0).to(dev)
        fkw = dict(token=token, token_len=torch.tensor([token.shape[1]], dtype=torch.int32), streaming=False, finalize=True)
        if z:
            fkw.update(prompt_token=mi['flow_prompt_speech_token'], prompt_token_len=mi['flow_prompt_speech_token_len'],
                       prompt_feat=mi['prompt_speech_feat'], prompt_feat_len=mi['prompt_speech_feat_len'], embedding=mi['flow_embedding'])
        else:
            fkw.update(embedding=mi['flow_embedding'].unsqueeze(0))
        mel, _ = flow.inference(**fkw)
        if speed != 1.0:
            mel = resample_linear(mel, max(1, int(mel.shape[2] / speed)))
        wav, _ = hift.inference(speech_feat=mel)
        outs.append(wav.cpu())
    total = time.time() - start
    audio = sum(o.shape[-1] for o in outs) / 24000
    logger.info('batched inference done: %d requests, total %.2fs, TPS %.2f, RTF %.4f', n, total,
                sum(len(t) for t in toks) / llm_time if llm_time > 0 else 0.0, total / audio if audio else 0.0)
    return outs


def inference_zero_shot(model_manager, tts_text, prompt_text, prompt_audio, prompt_sample_rate, speed=1.0):
    if not model_manager.is_loaded:
        raise ValueError('models are not loaded')
    try:
        fe = model_manager.frontend
        p_text = fe.text_normalize(prompt_text, split=False, text_frontend=True)
        t_text = fe.text_normalize(tts_text, split=True, text_frontend=True)
        model_input = fe.frontend_zero_shot(t_text[0], p_text, (prompt_audio, prompt_sample_rate), model_manager.configs['sample_rate'],
                                            zero_shot_spk_id='')
        return _synthesize(model_manager, model_input, speed, zero_shot=True)
    except Exception as e:
        raise ValueError('zero-shot inference failed: %s' % e)


def inference_tts(model_manager, text, spk_id, speed=1.0):
    if not model_manager.is_loaded:
        raise ValueError('models are not loaded')
    try:
        fe = model_manager.frontend
        t_text = fe.text_normalize(text, split=True, text_frontend=True)
        model_input = fe.frontend_sft(t_text[0], spk_id)
        return _synthesize(model_manager, model_input, speed, zero_shot=False)
    except Exception as e:
        raise ValueError('TTS inference failed: %s' % e)


_SEGMENT_MARKS = frozenset('。！？；，、.!?;,')


def split_text_by_punctuation(text, max_length=50, min_length=10):
    """Long-text segmentation of the reference (:263-315): texts up to `max_length` stay whole; otherwise cut after every punctuation
    mark once the running piece has `min_length` characters, glue a short tail to its predecessor, and fall back to fixed
    `max_length` slices when no mark produced a cut."""
    if len(text) <= max_length:
        return [text]
    pieces, start = [], 0
    for pos, ch in enumerate(text):
        if ch in _SEGMENT_MARKS and pos + 1 - start >= min_length:
            pieces.append(text[start:pos + 1])
            start = pos + 1
    tail = text[start:]
    if tail:
        if len(tail) < min_length and pieces:
            pieces[-1] += tail
        else:
            pieces.append(tail)
    if len(pieces) == 1 and len(pieces[0]) > max_length:
        pieces = [text[i:i + max_length] for i in range(0, len(text), max_length)]
    return pieces or [text]


def merge_short_segments(segments, min_length=5):
    """(:318-354) a piece shorter than `min_length` absorbs its successors until it is long enough; a short last piece joins the one
    before it"""
    out, cur = [], None
    for seg in segments:
        if cur is None:
            cur = seg
        elif len(cur) < min_length:
            cur += seg
        else:
            out.append(cur)
            cur = seg
    if cur:
        if len(cur) < min_length and out:
            out[-1] += cur
        else:
            out.append(cur)
    return out


def inference_tts_with_segmentation(model_manager, text, spk_id, max_length=30, min_length=10, last_prompt=True, speed=1.0):
    """(:357-452) piece-wise synthesis of a long text: the first piece (every piece when `last_prompt` is False) is plain speaker TTS,
    each later piece is zero-shot with the previous piece's text and audio as its prompt; pieces are joined with 50-150 ms of silence
    drawn from Python's `random` like the reference's."""
    import random
    segments = merge_short_segments(split_text_by_punctuation(text, max_length, min_length), min_length)
    if len(segments) == 1:
        return inference_tts(model_manager, text, spk_id, speed=speed)
    sr = model_manager.configs['sample_rate']
    parts, prev_text, prev_audio = [], None, None
    for i, seg in enumerate(segments):
        try:
            if i == 0 or not last_prompt:
                audio = inference_tts(model_manager, seg, spk_id, speed=speed)
            else:
                audio = inference_zero_shot(model_manager, seg, prev_text, prev_audio, 24000, speed=speed)
        except Exception as e:
            raise ValueError('segment %d failed: %s' % (i + 1, e))
        prev_text, prev_audio = seg, audio
        if parts:
            gap = list(audio.shape)
            gap[-1] = int(random.uniform(50, 150) * sr / 1000)
            parts.append(torch.zeros(gap, dtype=audio.dtype, device=audio.device))
        parts.append(audio)
    return torch.cat(parts, dim=-1)


def resolve_tts_request(model_manager, text, speaker_id):
    """the checks text_to_speech makes before it synthesises (infer_speech_model.py:743-780): models loaded, text not empty, a known speaker —
    or the first available one when the request names none.  Returns the speaker id to use; raises ValueError otherwise."""
    if not model_manager.is_loaded:
        raise ValueError('models are not loaded')
    if not text or not text.strip():
        raise ValueError('text is empty')
    speakers = [str(s['speaker_id'] if isinstance(s, dict) else s) for s in model_manager.get_available_speakers()]
    if speaker_id:
        if speakers and speaker_id not in speakers:
            raise ValueError('invalid speaker_id %s; available: %s' % (speaker_id, speakers))
        return speaker_id
    if speakers:
        return speakers[0]                                     # (:771-777) default: the first available speaker
    raise ValueError('no speaker available')


SEGMENTED_TEXT_CHARS = 5000        # longer texts take the segmented path (:782)


def text_to_speech(model_manager, text, speaker_id, speed=1.0):
    """-> {"output_audio": Tensor(1, L) cpu, "sample_rate", "format": "wav", "duration", "speaker_id", "segments_info"} (:743-820).
    Texts over 5000 characters take the segmented path (max 30 / min 10 characters per piece, speaker TTS for every piece, :782-800);
    `segments_info` is then {"total_segments", "segments"} and None otherwise, as in the reference."""
    try:
        speaker_id = resolve_tts_request(model_manager, text, speaker_id)
        segments_info = None
        if len(text) > SEGMENTED_TEXT_CHARS:
            audio = inference_tts_with_segmentation(model_manager, text, speaker_id, max_length=30, min_length=10, last_prompt=False, speed=speed)
            segs = merge_short_segments(split_text_by_punctuation(text, 30, 10), 10)
            segments_info = {'total_segments': len(segs), 'segments': segs}
        else:
            audio = inference_tts(model_manager, text, speaker_id, speed=speed)
        sr = model_manager.configs['sample_rate']
        return {'output_audio': audio, 'sample_rate': sr, 'format': 'wav', 'duration': audio.shape[-1] / sr, 'speaker_id': speaker_id,
                'segments_info': segments_info}
    except Exception as e:
        raise ValueError('TTS failed: %s' % e)
