"""KV-cache index arithmetic of the causal self-attention, on host integers.

Restates wan/modules/causal_model.py:202-236.  The reference keeps
`global_end_index` / `local_end_index` as device tensors and reads them back with `.item()` at
least twice per layer per forward; here the pipeline's integers drive the plan and the device
tensors are only kept up to date for schema compatibility.
"""
from __future__ import annotations

from dataclasses import dataclass


@dataclass(frozen=True)
class CachePlan:
    evict: int        # tokens dropped from the window before writing (0 = no roll)
    keep: int         # tokens moved from [sink+evict, sink+evict+keep) to [sink, sink+keep)
    sink: int         # sink tokens that never move
    local_end: int    # new local_end_index
    global_end: int   # new global_end_index
    write_start: int  # new K/V rows go to [write_start, local_end)
    attn_start: int   # attention reads rows [attn_start, local_end)


def plan_cache_update(local_end: int, global_end: int, current_start: int, num_new: int, capacity: int,
                      local_attn_size: int, sink_tokens: int, max_attention_size: int) -> CachePlan:
    """One self-attention call's cache update.

    * re-running a chunk (current_end == global_end) overwrites its slots in place;
    * rolling mode (local_attn_size != -1): on overflow the oldest non-sink tokens are evicted;
    * global mode overflow is an error (the reference fails with a slice-shape RuntimeError,
      causal_model.py:228)."""
    current_end = current_start + num_new
    evict = keep = 0
    if local_attn_size != -1 and current_end > global_end and num_new + local_end > capacity:
        evict = num_new + local_end - capacity
        keep = local_end - evict - sink_tokens
        new_local_end = local_end + current_end - global_end - evict
    else:
        new_local_end = local_end + current_end - global_end
    write_start = new_local_end - num_new
    if keep < 0 or write_start < 0 or new_local_end > capacity:
        raise RuntimeError(
            f"KV cache overflow: writing tokens [{write_start}, {new_local_end}) into a cache of {capacity} "
            f"(local_end={local_end}, global_end={global_end}, current_start={current_start}, new={num_new}, "
            f"local_attn_size={local_attn_size})")
    return CachePlan(evict=evict, keep=keep, sink=sink_tokens, local_end=new_local_end, global_end=current_end,
                     write_start=write_start, attn_start=max(0, new_local_end - max_attention_size))
