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


def worker_process_tts_batched(num_workers_gpu, task_queue, result_dict, worker_id, frontend_factory=None, max_batch=8):
    """Same wire format as worker_process_tts, with real batching: whatever `tts` / `zero_shot` tasks are already waiting
    (up to `max_batch`, same sampling parameters) are decoded together by `synthesize_many`; everything else is served one by one."""
    import queue as _queue
    os.environ['CUDA_VISIBLE_DEVICES'] = str(worker_id % num_workers_gpu)
    from .model_manager import HvxModelManager, synthesize_many, text_to_speech, inference_zero_shot
    from .sampling import ras_sampling

    mm = HvxModelManager(frontend_factory=frontend_factory)
    mm.load_models(argparse.Namespace(config=os.getenv('TTS_CONFIG'), model_dir=os.getenv('TTS_MODEL_DIR'), bf16=_env_flag('TTS_BF_16'),
                                      fp16=_env_flag('TTS_FP_16'), cpu=_env_flag('TTS_CPU', False)))
    normalise = _text_normaliser()                  # the same normaliser as worker_process_tts, in the batched and the one-by-one path
    pending = []
    stop = False
    while not stop or pending:                      # tasks deferred behind a load_pt / other sampling parameters are served before the exit
        if not pending:
            pending.append(task_queue.get())
        while not stop and len(pending) < max_batch:
            try:
                pending.append(task_queue.get_nowait())
            except _queue.Empty:
                break
        if any(t is None for t in pending):
            stop = True
            pending = [t for t in pending if t is not None]
        # the leading run of batchable tasks is decoded together; what follows the first task that cannot join (a load_pt, other sampling
        # parameters) waits for the next round, so arrival order around a hot swap is what the reference's one-by-one loop gives
        batch, rest, pending = group_batchable(pending)
        if len(batch) > 1:
            try:
                apply_extra_params(mm, batch[0], ras_sampling)
                fe = mm.frontend
                inputs, zs = [], []
                for t in batch:
                    if t['task_type'] == 'tts':
                        inputs.append(fe.frontend_sft(fe.text_normalize(normalise(t['text']), split=True, text_frontend=True)[0], t['speaker_id']))
                        zs.append(False)
                    else:
                        p_text = fe.text_normalize(normalise(t.get('prompt_text', '')), split=False, text_frontend=True)
                        inputs.append(fe.frontend_zero_shot(fe.text_normalize(normalise(t['tts_text']), split=True, text_frontend=True)[0], p_text,
                                                            (t['prompt_audio'], t['prompt_sample_rate']), mm.configs['sample_rate'], zero_shot_spk_id=''))
                        zs.append(True)
                speeds = [float(t.get('extra_params', {}).get('speed', 1.0)) for t in batch]
                sr = mm.configs['sample_rate']
                for t, out in zip(batch, synthesize_many(mm, inputs, zs, speeds=speeds)):
                    result_dict[t['id']] = {'output_audio': out, 'sample_rate': sr, 'format': t.get('output_format', 'wav'), 'duration': out.shape[-1] / sr}
            except Exception as e:
                logger.error('[TTS Worker-%d] batch error: %s', worker_id, e)
                for t in batch:
                    result_dict[t['id']] = {'error': str(e)}
        else:
            rest = batch + rest
        for t in rest:
            try:
                speed = apply_extra_params(mm, t, ras_sampling)
                if t['task_type'] == 'zero_shot':
                    out = inference_zero_shot(mm, normalise(t['tts_text']), normalise(t.get('prompt_text', '')), t['prompt_audio'], t['prompt_sample_rate'], speed=speed)
                    sr = mm.configs['sample_rate']
                    result = {'output_audio': out, 'sample_rate': sr, 'format': t.get('output_format', 'wav'), 'duration': out.shape[-1] / sr}
                elif t['task_type'] == 'tts':
                    result = text_to_speech(mm, normalise(t['text']), t['speaker_id'], speed=speed)
                elif t['task_type'] == 'load_pt':
                    result = mm.load_pt(t['llm_pt'], t['flow_pt'])
                else:
                    result = {'error': 'unknown task_type %r' % (t['task_type'],)}
            except Exception as e:
                logger.error('[TTS Worker-%d] Error: %s', worker_id, e)
                result = {'error': str(e)}
            result_dict[t['id']] = result


def group_batchable(tasks):
    """-> (batch, singles, later): `batch` = the leading run of synthesis tasks with the same sampling parameters (llm.sampling /
    inference_head_num are per-model state, server/worker.py:57-65); when the head of the queue cannot be batched (load_pt, unknown type) it
    is returned alone in `singles`; `later` = everything behind the first task that does not join, in arrival order — FIFO is kept, a
    load_pt is never overtaken by requests that arrived after it."""
    def key(t):
        ep = t.get('extra_params') or {}
        return tuple(ep.get(k) for k in ('top_p', 'top_k', 'win_size', 'tau_r', 'inference_head_num'))
    if not tasks:
        return [], [], []
    if tasks[0].get('task_type') not in ('tts', 'zero_shot'):
        return [], [tasks[0]], list(tasks[1:])
    k0 = key(tasks[0])
    n = 1
    while n < len(tasks) and tasks[n].get('task_type') in ('tts', 'zero_shot') and key(tasks[n]) == k0:
        n += 1
    return list(tasks[:n]), [], list(tasks[n:])


def apply_extra_params(mm, task, ras_sampling):
    """per-request knobs, exactly as server/worker.py:57-65 applies them; returns the speed"""
    ep = task.get('extra_params')
    if not ep:
        return 1.0
    mm.models['llm'].sampling = partial(ras_sampling, top_p=ep['top_p'], top_k=ep['top_k'], win_size=ep['win_size'], tau_r=ep['tau_r'])
    mm.models['llm'].inference_head_num = ep['inference_head_num']
    return float(ep.get('speed', 1.0))
