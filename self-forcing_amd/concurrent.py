"""Several rollouts in flight on ONE GPU, each on its own HIP stream.

Why: at batch 1 the big kernels of a rollout are single-round launches that cannot use the whole
chip (self-attention: 228 workgroups for 256 CUs; GEMM tails), and every forward has short
serial sections.  Prompts are independent (the path shards by prompt), the weights are shared
(2.84 GB) and a KV cache costs 6 GB of 288 GB, so a second rollout on a second stream simply fills
the idle CUs.  One Python thread per stream drives its own `CausalInferencePipeline`; the C-ABI
calls release the GIL while they enqueue kernels.
"""
from __future__ import annotations

import queue
import threading
from typing import Callable, List, Sequence

import torch

from .pipeline import CausalInferencePipeline
from .wan_wrapper import WanDiffusionWrapper


class RolloutPool:
    def __init__(self, args, device, generator: WanDiffusionWrapper, make_text_encoder: Callable[[], object],
                 make_vae: Callable[[], object], streams: int = 2):
        self.device = torch.device(device)
        self.pipes: List[CausalInferencePipeline] = []
        self.streams: List[torch.cuda.Stream] = []
        for i in range(max(1, streams)):
            gen = generator if i == 0 else generator.share()
            self.pipes.append(CausalInferencePipeline(args, device, generator=gen, text_encoder=make_text_encoder(), vae=make_vae()))
            self.streams.append(torch.cuda.Stream(device=self.device))

    def run_each(self, fn: Callable[[CausalInferencePipeline], object]) -> list:
        """fn(pipeline) once on EVERY stream's pipeline, concurrently (warm-up: cache and workspace
        allocation happen on first use)."""
        out = [None] * len(self.pipes)
        errors: list = []

        def worker(k):
            try:
                torch.cuda.set_device(self.device)
                with torch.cuda.stream(self.streams[k]), torch.no_grad():
                    out[k] = fn(self.pipes[k])
                self.streams[k].synchronize()
            except BaseException as e:  # noqa: BLE001
                errors.append(e)

        threads = [threading.Thread(target=worker, args=(k,)) for k in range(len(self.pipes))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
        return out

    def run(self, jobs: Sequence, fn: Callable[[CausalInferencePipeline, object], object]) -> list:
        """Apply fn(pipeline, job) to every job; jobs are pulled from one queue by the stream workers.
        Returns the results in job order.  Synchronises all streams before returning."""
        q: "queue.SimpleQueue" = queue.SimpleQueue()
        for i, j in enumerate(jobs):
            q.put((i, j))
        out = [None] * len(jobs)
        errors: list = []

        def worker(pipe, stream):
            try:
                torch.cuda.set_device(self.device)
                with torch.cuda.stream(stream), torch.no_grad():
                    while True:
                        try:
                            i, job = q.get_nowait()
                        except queue.Empty:
                            break
                        out[i] = fn(pipe, job)
                stream.synchronize()
            except BaseException as e:  # noqa: BLE001 - re-raised in the caller's thread
                errors.append(e)

        if len(self.pipes) == 1:
            worker(self.pipes[0], self.streams[0])
        else:
            threads = [threading.Thread(target=worker, args=(p, s)) for p, s in zip(self.pipes, self.streams)]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
        if errors:
            raise errors[0]
        return out
