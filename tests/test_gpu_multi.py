"""More than one GPU (BASELINE configs[3] / [4] are 8-GPU configurations; the driver's SCALE run launches `bench.py --gpus N` under torch.distributed.run).
The builder's boxes have ONE GPU, so this file skips there; on a node with >= 2 devices it is the first thing that must pass: two ranks over RCCL
("nccl" on ROCm), every rank its own GPU, the deal / hand-off / statistics path of bench.py with the real (toy-sized) pipeline."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(gpus, extra=()):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(gpus), '--tiny', '--steps', '2', '--warmup', '1', '--chars', '24', '--no-cpu-baseline',
           '--no-fp32-mode', '--no-extras'] + list(extra)
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_two_ranks_over_rccl_when_two_gpus_are_visible():
    if torch.cuda.device_count() < 2:
        pytest.skip('one GPU visible: the 2-rank RCCL path is covered on CPU by tests/test_host_cpu.py (gloo, world size 2) only')
    one = _bench(1)
    two = _bench(2)
    assert two['n_gpus'] == 2 and two['scaling'] == 'weak' and two['config']['global_batch'] == 2 * one['config']['global_batch']
    # pinned generation length: every utterance emits int(24 * 5.5) tokens, so the whole-job token count doubles exactly
    tok1 = one['value'] * one['ms_per_step'] * one['steps'] / 1e3
    tok2 = two['value'] * two['ms_per_step'] * two['steps'] / 1e3
    assert abs(tok2 - 2 * tok1) < 0.01 * tok2, (tok1, tok2)
    assert two['config']['parallelism'] == 'utterance-dp2'


def test_bench_single_rank_tiny_line_has_the_contract_keys():
    d = _bench(1)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'roofline'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 2 and d['value'] > 0 and 'TINY' in d['config']['workload']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in d['roofline'], k
