"""Utterance-level data parallelism across the GPUs of one node (one process per GPU, like the reference's one worker
process per GPU: server/worker.py:31, 122-127).

Utterances are independent end to end, so ranks never exchange anything on the data path.  The only collective is the
final hand-off of finished waveforms to rank 0: one small all_gather of the sample counts, then a *grouped* set of
point-to-point transfers (`batch_isend_irecv` -> ncclGroupStart/ncclSend|ncclRecv/ncclGroupEnd on RCCL), which fans in
over the xGMI links in parallel instead of serialising through a ring.  No all-reduce anywhere.
Seeds are the global utterance indices, so the audio does not depend on the world size.
"""
from typing import List, Sequence

import torch
import torch.distributed as dist


def shard_by_cost(costs: Sequence[float], world: int) -> List[List[int]]:
    """Longest-processing-time-first assignment of utterance indices to ranks (deterministic)."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    loads = [0.0] * world
    shards: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (loads[k], k))
        shards[r].append(i)
        loads[r] += costs[i]
    return [sorted(s) for s in shards]


def acoustic_cost(frames: float) -> float:
    """Work of the flow decoder + vocoder for one stream of `frames` mel frames, in flops: DiT Linears 7.56 GF and vocoder 0.672 GF per frame,
    DiT attention 1.80 MF per frame squared — the shard cost of BASELINE configs[4] (mixed lengths: the quadratic term makes a long stream worth more than
    its length)."""
    return (7.56e9 + 0.672e9) * frames + 1.80e6 * frames * frames


class Handoff:
    """Hands a rank's finished waveforms to rank `dst` in rounds of `per_round` utterances.  gather_waveforms is a collective, so every rank
    must enter it the same number of times although a longest-first deal gives the ranks different utterance counts: the number of rounds
    is fixed by the largest shard (every rank knows the whole deal), `push` enters one round whenever `per_round` waveforms are ready and
    `finish` enters the remaining ones (with whatever is left, possibly nothing)."""

    def __init__(self, shards: List[List[int]], per_round: int, dst: int = 0, group=None, keep: bool = True):
        self.per_round = max(1, int(per_round))
        self.rounds = max((len(s) + self.per_round - 1) // self.per_round for s in shards) if shards else 0
        self.dst, self.group = dst, group
        self.done = 0
        self.ready: List = []
        self.received = {}
        self.keep = keep                # False: only the last round's waveforms stay referenced (a long job would otherwise pile up on `dst`)
        self.n_received = 0

    def _round(self):
        ids, wavs = [g for g, _ in self.ready], [w for _, w in self.ready]
        got = gather_waveforms(wavs, ids, dst=self.dst, group=self.group)
        self.n_received += len(got)
        if self.keep:
            self.received.update(got)
        else:
            self.received = got
        self.ready = []
        self.done += 1

    def push(self, global_id: int, wav: torch.Tensor):
        self.ready.append((global_id, wav))
        if len(self.ready) == self.per_round:
            self._round()

    def finish(self):
        while self.done < self.rounds:
            self._round()
        assert not self.ready
        return self.received


def gather_waveforms(wavs: List[torch.Tensor], global_ids: List[int], dst: int = 0, group=None, shortcut_single: bool = True):
    """Every rank passes its finished waveforms (1-D float32 tensors on its device) with their global utterance ids.
    Rank `dst` returns {global_id: waveform}; the others return {}.  Works on RCCL ("nccl") and gloo.
    `shortcut_single=False` runs the collectives even in a group of one (the 1-GPU RCCL test)."""
    if not dist.is_available() or not dist.is_initialized() or (shortcut_single and dist.get_world_size(group) == 1):
        return dict(zip(global_ids, wavs))
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = wavs[0].device if wavs else torch.device('cuda' if dist.get_backend(group) == 'nccl' else 'cpu')
    n_local = torch.tensor([len(wavs)], dtype=torch.int64, device=dev)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    counts = [int(c.item()) for c in counts]
    cap = max(counts) if counts else 0
    meta = torch.full((max(cap, 1), 2), -1, dtype=torch.int64, device=dev)        # (global id, samples)
    for i, (gid, w) in enumerate(zip(global_ids, wavs)):
        meta[i, 0] = gid
        meta[i, 1] = w.numel()
    metas = [torch.empty_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    ops, out = [], {}
    if rank == dst:
        recv = {}
        for r in range(world):
            for i in range(counts[r]):
                gid, n = int(metas[r][i, 0]), int(metas[r][i, 1])
                if r == dst:
                    out[gid] = wavs[i]
                elif n > 0:
                    buf = torch.empty(n, dtype=torch.float32, device=dev)
                    recv[gid] = buf
                    ops.append(dist.P2POp(dist.irecv, buf, r, group=group))
                else:
                    out[gid] = torch.empty(0, dtype=torch.float32, device=dev)
        out.update(recv)
    else:
        for w in wavs:
            if w.numel() > 0:
                ops.append(dist.P2POp(dist.isend, w.contiguous(), dst, group=group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return out if rank == dst else {}
