"""The multi-GPU protocol of this path: prompt-sharded replicas, nothing on the data path.

The reference shards prompts over ranks and otherwise lets the ranks run alone (inference.py:39-50: `init_process_group`,
`set_seed(seed + rank)`; :96-107: `DistributedSampler(shuffle=False, drop_last=True)`, one `dist.barrier()`).  What
crosses GPUs here is exactly that plus what a benchmark needs to be believed:

    1. `init`            one process per GPU, `backend="nccl"` (= RCCL over xGMI on ROCm) or "gloo" (CPU tests), an
                         explicit collective timeout;
    2. `check_replicas`  every rank seeds the same weights; a checksum MIN / MAX all-reduce proves the replicas equal;
    3. `timed`           barrier -> rank-local work -> barrier, elapsed seconds MAX-reduced over ranks;
    4. `gather`          per-rank (units, seconds) to rank 0;
    5. `finish`          one last barrier with the ranks still together, then the group is destroyed -- rank 0 runs its
                         single-GPU report legs AFTER that, alone (a rank parked in an RCCL barrier keeps a spinning
                         kernel on its GPU for as long as it waits).

`bench.py` and the world-size-2 gloo tests call THIS code (tests/test_host_logic.py); nothing here touches a GPU
unless the caller passes a CUDA device.
"""
from __future__ import annotations

import datetime
import os
import subprocess
import sys
import time
from typing import Callable, List, Optional, Sequence

import torch


def env_rank_world():
    """(rank, local_rank, world) from the torch.distributed.run environment; (0, 0, 1) outside it."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def launched_by_torchrun() -> bool:
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def free_port() -> int:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(script: str, argv: Sequence[str], nproc: int, port: Optional[int] = None, timeout: Optional[float] = None) -> int:
    """Start `nproc` rank processes of `script` as CHILDREN through torch.distributed.run (one per GPU, rendezvous on
    127.0.0.1) and return the launcher's exit code; rank 0's stdout (the JSON line) passes through.  Must be called
    before this process makes any GPU call: the children are fresh processes, the parent only waits (never an exec of
    a process that has initialised the GPU)."""
    port = port or int(os.environ.get("MASTER_PORT", 0)) or free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), script, *argv]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("ROC_SIGNAL_POOL_SIZE", "1024")         # (see _lib.py: keeps a runtime helper thread from spinning)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(1, nproc))))
    return subprocess.run(cmd, env=env, timeout=timeout).returncode


class RankGroup:
    """The protocol above over torch.distributed; outside torch.distributed.run (world == 1) every method is the identity
    (no process group)."""

    def __init__(self, backend: Optional[str] = None, device: Optional[torch.device] = None, timeout_s: float = 1800.0):
        self.rank, self.local_rank, self.world = env_rank_world()
        self.device = device if device is not None else torch.device("cpu")
        self.backend = backend or ("nccl" if self.device.type == "cuda" else "gloo")
        self.dist = None
        # a process group whenever torch.distributed.run started us -- also with ONE rank, so that the RCCL code path
        # (init with device_id, all_reduce MIN / MAX, barrier, all_gather, destroy) can be exercised on a one-GPU box
        if self.world > 1 or launched_by_torchrun():
            import torch.distributed as dist
            kw = {"device_id": self.device} if self.backend == "nccl" else {}
            dist.init_process_group(backend=self.backend, timeout=datetime.timedelta(seconds=timeout_s), **kw)
            self.dist = dist

    # tensors of the collectives live where the backend can reach them
    def _t(self, values, dtype=torch.float64):
        return torch.tensor(values, dtype=dtype, device=self.device if self.backend == "nccl" else "cpu")

    def check_replicas(self, checksum: float) -> float:
        """Raises unless every rank holds the same checksum (bit-equal as float64); returns it."""
        if self.dist is not None:
            lo, hi = self._t([checksum]), self._t([checksum])
            self.dist.all_reduce(lo, op=self.dist.ReduceOp.MIN)
            self.dist.all_reduce(hi, op=self.dist.ReduceOp.MAX)
            if lo.item() != hi.item():
                raise RuntimeError(f"weight replicas differ across ranks: checksum min {lo.item()!r} max {hi.item()!r}")
        return float(checksum)

    def barrier(self) -> None:
        if self.dist is not None:
            self.dist.barrier()

    def _sync(self) -> None:
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)

    def timed(self, work: Callable[[], object]):
        """barrier + device sync | work() | device sync + barrier + device sync; returns (MAX-over-ranks seconds, this
        rank's seconds, work's result)."""
        self._sync()
        self.barrier()
        self._sync()
        t0 = time.perf_counter()
        res = work()
        self._sync()
        local = time.perf_counter() - t0
        self.barrier()
        self._sync()
        elapsed = time.perf_counter() - t0
        if self.dist is not None:
            tt = self._t([elapsed])
            self.dist.all_reduce(tt, op=self.dist.ReduceOp.MAX)
            elapsed = tt.item()
        return elapsed, local, res

    def gather(self, values: Sequence[float]) -> List[List[float]]:
        """Every rank's `values` (same length everywhere), indexed by rank; complete on every rank."""
        if self.dist is None:
            return [list(map(float, values))]
        mine = self._t(list(values))
        out = [torch.zeros_like(mine) for _ in range(self.world)]
        self.dist.all_gather(out, mine)
        return [o.tolist() for o in out]

    def finish(self) -> None:
        """Last collective: one barrier while all ranks are still here, then tear the group down."""
        if self.dist is not None:
            self.dist.barrier()
            self._sync()
            self.dist.destroy_process_group()
            self.dist = None


def selftest(steps: int = 3, warmup: int = 1) -> dict:
    """The whole protocol on CPU (gloo), no GPU work: seeded reduced-shape weights as the replica, a rank-dependent
    sleep as the 'rollout'.  Returns the report dict (rank 0 prints it); used by `bench.py --dist-selftest`."""
    from .sharding import shard_indices
    from .weights import WAN_REDUCED, synth_state_dict
    grp = RankGroup(backend="gloo")
    sd = synth_state_dict(WAN_REDUCED, seed=0)
    grp.check_replicas(float(torch.stack([v.double().sum() for v in sd.values()]).sum()))
    total = warmup + steps
    idx = shard_indices(total * grp.world, grp.rank, grp.world)

    def work():
        time.sleep(0.05 * (grp.rank + 1) * steps)          # the slowest rank sets the time
        return len(idx) - warmup

    elapsed, local, done = grp.timed(work)
    per_rank = grp.gather([float(done), local] + [float(i) for i in idx])
    grp.finish()
    return {"metric": "dist-selftest (gloo, no GPU work)", "n_gpus": grp.world, "steps": steps, "warmup": warmup,
            "elapsed_s": elapsed, "ms_per_step": 1e3 * elapsed / steps, "units_per_rank": [int(r[0]) for r in per_rank],
            "seconds_per_rank": [r[1] for r in per_rank], "prompt_indices_per_rank": [[int(i) for i in r[2:]] for r in per_rank],
            "rank": grp.rank, "backend": "gloo"}
