"""Prompt sharding for multi-GPU inference: the only parallelism of this path.

Restates the semantics of `DistributedSampler(dataset, shuffle=False, drop_last=True)` +
`DataLoader(batch_size=1)` as used by the reference (inference.py:96-100): with P prompts and W
ranks only the first W * floor(P / W) prompts are used and rank r takes r, r + W, r + 2W, ...
No tensor crosses GPUs; RCCL is used for the start-up weight check and barriers only."""
from __future__ import annotations

from typing import List, Sequence


def shard_indices(num_prompts: int, rank: int, world_size: int) -> List[int]:
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError(f"bad rank/world_size {rank}/{world_size}")
    per_rank = num_prompts // world_size
    return [rank + world_size * i for i in range(per_rank)]


def shard(prompts: Sequence[str], rank: int, world_size: int) -> List[str]:
    return [prompts[i] for i in shard_indices(len(prompts), rank, world_size)]


def read_prompts(path: str, first_n: int = 0) -> List[str]:
    """One prompt per line (utils/dataset.py:12-34 TextDataset); `first_n` = eval_first_n of
    configs/default_config.yaml:19-21."""
    with open(path, encoding="utf-8") as f:
        lines = [ln.strip() for ln in f if ln.strip()]
    return lines[:first_n] if first_n else lines
