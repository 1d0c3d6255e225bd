"""Rank-sharded loaders for the data-parallel trainers (SURVEY.md section 8(e) "Partitioning"; the reference has one loader on one device:
REF/train/visionlanguage_tasks/train_vqa.py:64-83, :219).

Rule: `--batch_size` stays the GLOBAL batch.  A run on N ranks draws, with the SAME random stream on every rank, the
very permutation a single process would draw (torch's RandomSampler algorithm and its place in the RNG call order are kept), cuts it into
global batches of `batch_size`, and rank r takes elements r, r + N, r + 2N ... of each global batch.  Consequences:
  * global batch k of an N-rank run is batch k of the single-GPU run with the same seed -- the update is the single-GPU update up to fp
    summation order, and steps per epoch (hence the reference's warm-up / decay schedule) are unchanged;
  * a last, partial global batch of G' examples gives ranks unequal shares b_r; the loss is a batch MEAN, so the average over ranks of the
    shard gradients equals the global gradient only if rank r's d(loss) is weighted by b_r * N / G'.  The loader hands that weight to
    `train_step` as `batch["dp_weight"]` (1.0 whenever the shares are equal);
  * G' < N leaves ranks without examples: they repeat one (index r mod G') with weight 0, so every rank still runs the step and takes part
    in the collectives while contributing exactly nothing;
  * evaluation shards are exact (no padding, no repeats): rank r scores its own examples and the trainer all-reduces the score sum.
`replicated()` switches a loader to whole global batches on every rank: the Fisher pass (REF/cl_algorithms/ewc.py:56-68 accumulates
gradients across batches IN ORDER, so it is not shardable without changing the result).
"""
from __future__ import annotations

import contextlib
from collections import deque
from typing import Iterator, List, Sequence, Tuple

import torch
from torch.utils.data import DataLoader, Sampler


def dp_rank_world(group=None) -> Tuple[int, int]:
    """(rank, world size) of the data-parallel job; (0, 1) when torch.distributed is not initialised."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def shard_of(indices: Sequence[int], rank: int, world: int, pad: bool = True) -> Tuple[List[int], float]:
    """Rank `rank`'s strided share of one global batch and the weight its d(loss) must carry (see the module docstring)."""
    indices = list(indices)
    if world == 1:
        return indices, 1.0
    mine = indices[rank::world]
    if mine:
        return mine, len(mine) * world / len(indices)
    if pad and indices:
        return [indices[rank % len(indices)]], 0.0
    return [], 0.0


class ShardedBatchSampler(Sampler):
    """batch_sampler of a `ShardedDataLoader`."""

    def __init__(self, dataset, global_batch: int, shuffle: bool, rank: int, world: int, pad: bool = True, generator=None):
        if global_batch % world:
            raise ValueError(f"data parallel: a global batch of {global_batch} examples does not divide over {world} ranks "
                             "(--batch_size is the GLOBAL batch; NLVR2 / VCR loaders use a half / a quarter of it)")
        self.dataset, self.global_batch, self.shuffle = dataset, int(global_batch), bool(shuffle)
        self.rank, self.world, self.pad, self.generator = rank, world, pad, generator
        self.replicated = False
        self.weights = deque()          # one entry per yielded batch, consumed by ShardedDataLoader in the same order

    def __len__(self) -> int:
        return (len(self.dataset) + self.global_batch - 1) // self.global_batch

    def global_batches(self) -> Iterator[List[int]]:
        n = len(self.dataset)
        if self.shuffle:             # torch.utils.data.RandomSampler.__iter__, so a seeded run sees the batches the single-GPU run sees
            gen = self.generator
            if gen is None:
                seed = int(torch.empty((), dtype=torch.int64).random_().item())
                gen = torch.Generator()
                gen.manual_seed(seed)
            order = torch.randperm(n, generator=gen).tolist()
        else:
            order = list(range(n))
        for k in range(0, n, self.global_batch):
            yield order[k:k + self.global_batch]

    def __iter__(self) -> Iterator[List[int]]:
        self.weights.clear()
        for glob in self.global_batches():
            if self.replicated:
                mine, w = glob, 1.0
            else:
                mine, w = shard_of(glob, self.rank, self.world, self.pad)
                if not mine:
                    continue
            # (w, examples of the global batch per rank): the second is what the half build's loss scale is chosen from under data parallelism --
            # the same number on every rank, also on one that only repeats an example with weight 0 (engine.begin_scaled_backward)
            self.weights.append((w, len(glob) / max(1, self.world) if not self.replicated else float(len(glob))))
            yield mine


class ShardedDataLoader(DataLoader):
    """A DataLoader over this rank's shard; every batch dict carries `dp_weight`.  `len()` = global steps per epoch on every rank."""

    def __init__(self, dataset, global_batch: int, shuffle: bool, collate_fn, num_workers: int = 0, rank: int = 0, world: int = 1, pad: bool = True):
        sampler = ShardedBatchSampler(dataset, global_batch, shuffle, rank, world, pad)
        super().__init__(dataset, batch_sampler=sampler, collate_fn=collate_fn, num_workers=num_workers)
        self.global_batch = int(global_batch)

    def __iter__(self):
        sampler = self.batch_sampler
        for batch in super().__iter__():
            w, rows = sampler.weights.popleft()
            if isinstance(batch, dict):
                batch["dp_weight"] = w
                batch["dp_rows"] = rows
            yield batch

    @contextlib.contextmanager
    def replicated(self):
        """Inside: every rank iterates the WHOLE global batches (what a single process would see), weight 1."""
        prev = self.batch_sampler.replicated
        self.batch_sampler.replicated = True
        try:
            yield self
        finally:
            self.batch_sampler.replicated = prev


def replicated(loader):
    """Context manager: `loader.replicated()` for a sharded loader, a no-op for a plain one."""
    return loader.replicated() if isinstance(loader, ShardedDataLoader) else contextlib.nullcontext(loader)
