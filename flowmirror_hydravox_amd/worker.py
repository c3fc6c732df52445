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


def worker_process_tts(num_workers_gpu, task_queue, result_dict, worker_id, frontend_factory=None):
    os.environ['CUDA_VISIBLE_DEVICES'] = str(worker_id % num_workers_gpu)
    from .model_manager import HvxModelManager, text_to_speech, inference_zero_shot
    from .sampling import ras_sampling

    model_manager = HvxModelManager(frontend_factory=frontend_factory)
    args = argparse.Namespace(config=os.getenv('TTS_CONFIG'), model_dir=os.getenv('TTS_MODEL_DIR'), bf16=_env_flag('TTS_BF_16'),
                              fp16=_env_flag('TTS_FP_16'), cpu=_env_flag('TTS_CPU', False))
    model_manager.load_models(args)
    try:
        from fmtn import create_default_tn
        tn = create_default_tn(verbose=True)
        normalise = tn.process_text
    except Exception:
        normalise = lambda s: s                                # noqa: E731
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
