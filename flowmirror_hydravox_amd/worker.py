"""worker_process_tts — same task / result protocol as the reference's GPU worker (server/worker.py:25-102).

One process per GPU (`CUDA_VISIBLE_DEVICES = worker_id % num_workers_gpu`, :31), tasks popped from a
`multiprocessing.Manager().Queue()`, results written to a `Manager().dict()` keyed by task id; per-request knobs are applied
by mutating the llm object (`llm.sampling = partial(ras_sampling, top_p=, top_k=, win_size=, tau_r=)`,
`llm.inference_head_num`, :57-65); errors never propagate, they become `{"error": str(e)}` (:98-100); `None` is the shutdown
sentinel (:67-68).  The proprietary `fmtn` text normaliser is optional here (its wheel is absent from the reference tree).
"""
import argparse
import logging
import os
from functools import partial

logger = logging.getLogger('hvx.worker')


def _env_flag(name, default=False):
    raw = os.getenv(name)
    if raw is None:
        return default
    return raw.strip().lower() in {'1', 'true', 'yes', 'on'}


def _text_normaliser():
    """the reference worker's `fmtn` normaliser (server/worker.py:46-52, 72-88); identity when the proprietary wheel is absent"""
    try:
        from fmtn import create_default_tn
        return create_default_tn(verbose=True).process_text
    except Exception:
        return lambda s: s


def worker_process_tts(num_workers_gpu, task_queue, result_dict, worker_id, frontend_factory=None):
    os.environ['CUDA_VISIBLE_DEVICES'] = str(worker_id % num_workers_gpu)
    from .model_manager import HvxModelManager, text_to_speech, inference_zero_shot
    from .sampling import ras_sampling

    model_manager = HvxModelManager(frontend_factory=frontend_factory)
    args = argparse.Namespace(config=os.getenv('TTS_CONFIG'), model_dir=os.getenv('TTS_MODEL_DIR'), bf16=_env_flag('TTS_BF_16'),
                              fp16=_env_flag('TTS_FP_16'), cpu=_env_flag('TTS_CPU', False))
    model_manager.load_models(args)
    normalise = _text_normaliser()
    while True:
        task = task_queue.get()
        if task is None:
            break
        speed = 1.0
        if 'extra_params' in task:
            ep = task['extra_params']
            model_manager.models['llm'].sampling = partial(ras_sampling, top_p=ep['top_p'], top_k=ep['top_k'], win_size=ep['win_size'],
                                                           tau_r=ep['tau_r'])
            model_manager.models['llm'].inference_head_num = ep['inference_head_num']
            speed = float(ep.get('speed', 1.0))
        try:
            task_type = task['task_type']
            if task_type == 'zero_shot':
                out = inference_zero_shot(model_manager, normalise(task['tts_text']), normalise(task.get('prompt_text', '')),
                                          task['prompt_audio'], task['prompt_sample_rate'], speed=speed)
                sr = model_manager.configs['sample_rate']
                result = {'output_audio': out, 'sample_rate': sr, 'format': task.get('output_format', 'wav'), 'duration': out.shape[-1] / sr}
            elif task_type == 'tts':
                result = text_to_speech(model_manager, normalise(task['text']), task['speaker_id'], speed=speed)
            elif task_type == 'load_pt':
                result = model_manager.load_pt(task['llm_pt'], task['flow_pt'])
            else:
                result = {'error': 'unknown task_type %r' % (task_type,)}
        except Exception as e:
            logger.error('[TTS Worker-%d] Error: %s', worker_id, e)
            result = {'error': str(e)}
        result_dict[task['id']] = result


def _is_segmented(task, normalise):
    """a `tts` task whose text takes the reference's segmented path (over 5000 characters, infer_speech_model.py:782-800): many short speaker-TTS
    calls concatenated — served one by one through text_to_speech, as an epoch boundary of the batched worker"""
    from .model_manager import SEGMENTED_TEXT_CHARS
    if task.get('task_type') != 'tts':
        return False
    try:
        return len(normalise(task.get('text') or '')) > SEGMENTED_TEXT_CHARS
    except Exception:
        return False


def _sampling_key(task):
    ep = task.get('extra_params') or {}
    return tuple(ep.get(k) for k in ('top_p', 'top_k', 'win_size', 'tau_r', 'inference_head_num'))


class _TaskSource:
    """The polling source HvxPipeline.serve reads (one EPOCH of the task queue): consecutive `tts` / `zero_shot` tasks with the same sampling
    parameters become utterances as they arrive and join the decode grid while earlier ones are still in flight.  The first task that cannot
    join — a `load_pt`, other sampling parameters (llm.sampling / inference_head_num are per-model state, server/worker.py:57-65), an unknown
    type, the shutdown sentinel — closes the epoch: it is kept in `carry` and served after everything accepted before it has finished, so
    arrival order around a hot swap is what the reference's one-by-one loop gives."""

    def __init__(self, mm, task_queue, result_dict, first, normalise, worker_id):
        self.mm, self.q, self.results, self.normalise, self.worker_id = mm, task_queue, result_dict, normalise, worker_id
        self.key = _sampling_key(first)
        self.head = first
        self.carry = None
        self.stop = False
        self.closed = False

    POLL_SECONDS = 0.1

    def unpoll(self, u):
        """an utterance the engine fetched while it was being cancelled: its task opens the next epoch instead of failing with this one"""
        self.carry, self.closed = u.tag, True

    def _utterance(self, t):
        from .pipeline import Utterance
        from .model_manager import resolve_tts_request
        fe, mm = self.mm.frontend, self.mm
        if t['task_type'] == 'tts':
            # the same checks and the same default speaker as the one-by-one path (model_manager.text_to_speech, infer_speech_model.py:743-780)
            try:
                text = self.normalise(t['text'])
                speaker = resolve_tts_request(mm, text, t.get('speaker_id'))
            except Exception as e:
                raise ValueError('TTS failed: %s' % e)
            mi = fe.frontend_sft(fe.text_normalize(text, split=True, text_frontend=True)[0], speaker)
            u = Utterance(text=mi['text'].reshape(-1), seed=None, embedding=mi['flow_embedding'].reshape(-1))
            u.speaker_id = speaker
        else:
            p_text = fe.text_normalize(self.normalise(t.get('prompt_text', '')), split=False, text_frontend=True)
            mi = fe.frontend_zero_shot(fe.text_normalize(self.normalise(t['tts_text']), split=True, text_frontend=True)[0], p_text,
                                       (t['prompt_audio'], t['prompt_sample_rate']), mm.configs['sample_rate'], zero_shot_spk_id='')
            u = Utterance(text=mi['text'].reshape(-1), seed=None, embedding=mi['flow_embedding'].reshape(-1),
                          prompt_text=mi['prompt_text'].reshape(-1), prompt_speech_token=mi['llm_prompt_speech_token'].reshape(-1),
                          prompt_feat=mi['prompt_speech_feat'].reshape(-1, mi['prompt_speech_feat'].shape[-1]),
                          flow_prompt_token=mi['flow_prompt_speech_token'].reshape(-1))
        u.seed = int(t['seed']) if t.get('seed') is not None else None       # (None: the LM draws one from the global generator)
        u.speed = float((t.get('extra_params') or {}).get('speed', 1.0))
        if u.speed <= 0:
            raise ValueError('Invalid speed: %s' % u.speed)
        u.tag = t
        return u

    def poll(self, block):
        import queue as _queue
        while True:
            if self.closed:
                raise StopIteration
            if self.head is not None:
                t, self.head = self.head, None
            else:
                try:
                    # bounded: the engine asks again (and looks at its cancellation flag) every time this returns None — a blocking get()
                    # would hold a cancelled engine, and the error replies of its requests, until the next task arrived
                    t = self.q.get(timeout=self.POLL_SECONDS) if block else self.q.get_nowait()
                except _queue.Empty:
                    return None
            if t is None:
                self.stop = self.closed = True
                raise StopIteration
            if t.get('task_type') not in ('tts', 'zero_shot') or _sampling_key(t) != self.key or _is_segmented(t, self.normalise):
                self.carry, self.closed = t, True
                raise StopIteration
            try:
                return self._utterance(t)
            except Exception as e:                     # a frontend error fails that request only
                logger.error('[TTS Worker-%d] Error: %s', self.worker_id, e)
                self.results[t['id']] = {'error': str(e)}


def worker_process_tts_batched(num_workers_gpu, task_queue, result_dict, worker_id, frontend_factory=None, lm_slots=None, acoustic_batch=4):
    """Same wire format as worker_process_tts, served by the continuous-batching engine (replaces the one-request-at-a-time
    loop of server/worker.py:54-102 and what server/router.py:144-156 feeds it): `tts` / `zero_shot` tasks join ONE decode grid as they arrive
    (HvxPipeline.serve -> HvxLLM.generate_stream: the LM weights are streamed once per step for every request in flight, a finished request's
    slot goes to the next waiting one), finished requests go to length-bucketed padded CFM solves and the vocoder beside the decode of the
    others, and every result is written to `result_dict` the moment its waveform is complete.  `load_pt` and a change of sampling parameters
    are epoch boundaries (FIFO: see _TaskSource).  What bench.py measures is this path."""
    os.environ['CUDA_VISIBLE_DEVICES'] = str(worker_id % num_workers_gpu)
    from .model_manager import HvxModelManager
    from .pipeline import HvxPipeline
    from .sampling import ras_sampling

    mm = HvxModelManager(frontend_factory=frontend_factory)
    mm.load_models(argparse.Namespace(config=os.getenv('TTS_CONFIG'), model_dir=os.getenv('TTS_MODEL_DIR'), bf16=_env_flag('TTS_BF_16'),
                                      fp16=_env_flag('TTS_FP_16'), cpu=_env_flag('TTS_CPU', False)))
    serve_queue(mm, task_queue, result_dict, worker_id, lm_slots=lm_slots, acoustic_batch=acoustic_batch)


def serve_queue(mm, task_queue, result_dict, worker_id=0, lm_slots=None, acoustic_batch=4, normalise=None):
    """the loop of worker_process_tts_batched over an already loaded model manager (tests drive this directly)"""
    from .pipeline import HvxPipeline
    from .sampling import ras_sampling
    normalise = normalise or _text_normaliser()
    pipe = HvxPipeline.from_models(mm.hvx_config, mm.models['llm'], mm.models['flow'], mm.models['hift'], acoustic_batch=acoustic_batch)
    lm_slots = lm_slots or mm.models['llm'].max_batch
    sr = mm.configs['sample_rate']
    carry = None
    while True:
        task, carry = (carry, None) if carry is not None else (task_queue.get(), None)
        if task is None:
            break
        if task.get('task_type') not in ('tts', 'zero_shot') or _is_segmented(task, normalise):
            try:
                if task.get('task_type') == 'load_pt':
                    result = mm.load_pt(task['llm_pt'], task['flow_pt'])
                elif task.get('task_type') == 'tts':               # segmented text: the one-by-one path, with this request's own knobs
                    from .model_manager import text_to_speech
                    result = text_to_speech(mm, normalise(task['text']), task.get('speaker_id'), speed=apply_extra_params(mm, task, ras_sampling))
                else:
                    result = {'error': 'unknown task_type %r' % (task.get('task_type'),)}
            except Exception as e:
                logger.error('[TTS Worker-%d] Error: %s', worker_id, e)
                result = {'error': str(e)}
            result_dict[task['id']] = result
            continue
        src = _TaskSource(mm, task_queue, result_dict, task, normalise, worker_id)
        in_flight = {}
        try:
            apply_extra_params(mm, task, ras_sampling)
            pipe.llm = mm.models['llm']
            for u, wav, toks in pipe.serve(_Tracked(src, in_flight), lm_slots=lm_slots, acoustic_batch=acoustic_batch, acoustic_min_batch=1):
                t = u.tag
                in_flight.pop(t['id'], None)
                if isinstance(wav, BaseException):
                    logger.error('[TTS Worker-%d] Error: %s', worker_id, wav)
                    result_dict[t['id']] = {'error': str(wav)}
                else:
                    out = wav.reshape(1, -1).cpu()
                    if t['task_type'] == 'tts':                 # the result keys of text_to_speech (infer_speech_model.py:802-812)
                        result_dict[t['id']] = {'output_audio': out, 'sample_rate': sr, 'format': 'wav', 'duration': out.shape[-1] / sr,
                                                'speaker_id': getattr(u, 'speaker_id', t.get('speaker_id')), 'segments_info': None}
                    else:                                      # ... and of the worker's zero_shot branch (server/worker.py:76-83)
                        result_dict[t['id']] = {'output_audio': out, 'sample_rate': sr, 'format': t.get('output_format', 'wav'), 'duration': out.shape[-1] / sr}
        except Exception as e:                         # an engine failure fails what was in flight, never the worker
            logger.error('[TTS Worker-%d] epoch error: %s', worker_id, e)
            for tid in list(in_flight):
                result_dict[tid] = {'error': str(e)}
        carry = src.carry
        if src.stop:
            break


class _Tracked:
    """notes which accepted tasks have no result yet (an engine failure must answer them)"""

    def __init__(self, src, in_flight):
        self.src, self.in_flight = src, in_flight

    def poll(self, block):
        u = self.src.poll(block)
        if u is not None:
            self.in_flight[u.tag['id']] = True
        return u

    def unpoll(self, u):
        self.in_flight.pop(u.tag['id'], None)
        self.src.unpoll(u)


def apply_extra_params(mm, task, ras_sampling):
    """per-request knobs, exactly as server/worker.py:57-65 applies them; returns the speed"""
    ep = task.get('extra_params')
    if not ep:
        return 1.0
    mm.models['llm'].sampling = partial(ras_sampling, top_p=ep['top_p'], top_k=ep['top_k'], win_size=ep['win_size'], tau_r=ep['tau_r'])
    mm.models['llm'].inference_head_num = ep['inference_head_num']
    return float(ep.get('speed', 1.0))
