"""Host-side logic of the multi-GPU path: how frames are sharded, and the few
collectives one VI call needs.  Device-independent on purpose (tensors may be CPU/gloo or
GPU/RCCL), so the N>1 path is covered by world-size-2 gloo tests on CPU
(tests/test_dist_cpu.py).

One process per GPU.  Frames are sharded BY SAMPLE on VIDEO boundaries: the per-video stick
statistics, gamma and the prior table then stay rank-local, and the only data-path collective
of an iteration is one all-reduce (sum, fp64) of the per-leaf statistics [S0 | S1 (| Q)].
"""
from __future__ import annotations

import os

import numpy as np


def world_size(group):
    if group is None:
        return 1
    import torch.distributed as dist
    return dist.get_world_size(group)


def is_distributed(group):
    """True when the collectives below must run: more than one rank -- or a group of ONE rank under
    BNPVI_DIST_FORCE_COLLECTIVES=1, the switch with which a one-GPU box runs every collective of the path through
    RCCL itself (tests/test_gpu_round3.py::test_rccl_collectives_on_a_single_rank_group); sums, gathers and
    broadcasts over one rank leave the results unchanged."""
    if group is None:
        return False
    return world_size(group) > 1 or os.environ.get("BNPVI_DIST_FORCE_COLLECTIVES") == "1"


def shard_videos(frames_per_video, world):
    """Contiguous split of the videos over `world` ranks, balanced by frame count.

    Returns a list of (video_lo, video_hi, frame_lo, frame_hi) per rank; every video belongs to
    exactly one rank (a rank may be empty when there are fewer videos than ranks)."""
    fpv = np.asarray(frames_per_video, dtype=np.int64)
    ends = np.cumsum(fpv)
    total = int(ends[-1]) if len(ends) else 0
    out, v0, f0 = [], 0, 0
    for r in range(world):
        target = total * (r + 1) // world            # ideal cumulative frame count after rank r
        if r == world - 1:
            v1 = len(fpv)
        else:
            # first video boundary at or past the target, but leave the cut where it is closer
            v1 = int(np.searchsorted(ends, target, side="left"))
            if v1 < len(fpv) and v1 >= v0:
                before = int(ends[v1 - 1]) if v1 > 0 else 0
                after = int(ends[v1])
                v1 = v1 + 1 if (after - target) < (target - before) else v1
            v1 = max(v1, v0)
        f1 = int(ends[v1 - 1]) if v1 > 0 else 0
        out.append((v0, v1, f0, f1))
        v0, f0 = v1, f1
    return out


def _host_staged(group):
    """gloo groups (CPU tests, one-GPU rehearsals) get host tensors; RCCL works on the device."""
    import torch.distributed as dist
    return dist.get_backend(group) == "gloo"


def allreduce_sum_(t, group):
    """In-place sum over ranks (the single per-iteration collective on the statistics buffer)."""
    if is_distributed(group):
        import torch.distributed as dist
        if t.is_cuda and _host_staged(group):
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def global_count(n_local, group, device):
    """(sum of n_local over ranks, exclusive prefix = global index of this rank's first frame)."""
    if not is_distributed(group):
        return int(n_local), 0
    import torch
    import torch.distributed as dist
    if _host_staged(group):
        device = "cpu"
    mine = torch.tensor([int(n_local)], dtype=torch.int64, device=device)
    allv = [torch.empty_like(mine) for _ in range(dist.get_world_size(group))]
    dist.all_gather(allv, mine, group=group)
    counts = [int(a.item()) for a in allv]
    return sum(counts), sum(counts[:dist.get_rank(group)])


def select_argmax(score, global_row, group, device):
    """Arg-max over ranks of (score, global row): the highest score wins, the LOWEST global row
    wins ties (np.argmax semantics of src/var_inf.py:30,35).  A rank without frames passes
    score = -1.  Returns (winning global row, rank that owns it)."""
    if not is_distributed(group):
        return int(global_row), 0
    import torch
    import torch.distributed as dist
    me = dist.get_rank(group)
    if _host_staged(group):
        device = "cpu"
    mine = torch.tensor([float(score), float(global_row), float(me)], dtype=torch.float64, device=device)
    allv = [torch.empty_like(mine) for _ in range(dist.get_world_size(group))]
    dist.all_gather(allv, mine, group=group)
    cand = sorted((-float(a[0]), float(a[1]), int(a[2])) for a in allv)
    _negs, grow, owner = cand[0]
    return int(grow), owner


def select_argmax_many(scores, global_rows, group):
    """`select_argmax` for J columns at once: scores [J] float64 and global_rows [J] int64 device tensors
    of this rank's winners.  Returns (winning global rows [J] int64, mask [J] of the columns this rank owns)
    on the tensors' device -- one all-gather of a [J, 2] table."""
    import torch
    if not is_distributed(group):
        return global_rows, torch.ones_like(global_rows, dtype=torch.bool)
    import torch.distributed as dist
    me = dist.get_rank(group)
    dev = scores.device
    mine = torch.stack([scores.to(torch.float64), global_rows.to(torch.float64)], dim=1)
    if _host_staged(group):
        mine = mine.cpu()
    allv = [torch.empty_like(mine) for _ in range(dist.get_world_size(group))]
    dist.all_gather(allv, mine.contiguous(), group=group)
    table = torch.stack(allv, dim=0)                                  # [W, J, 2]
    sc, rows = table[:, :, 0], table[:, :, 1]
    best = sc.max(dim=0, keepdim=True).values
    cand = torch.where(sc == best, rows, torch.full_like(rows, float("inf")))
    grow, owner = cand.min(dim=0)                                     # lowest global row among the best
    return grow.to(torch.int64).to(dev), (owner == me).to(dev)


def broadcast_from(t, owner, group):
    """Broadcast `t` from group rank `owner` (the chosen k-means centre row)."""
    if is_distributed(group):
        import torch.distributed as dist
        src = dist.get_global_rank(group, owner)
        if t.is_cuda and _host_staged(group):
            h = t.cpu()
            dist.broadcast(h, src=src, group=group)
            t.copy_(h)
        else:
            dist.broadcast(t, src=src, group=group)
    return t
