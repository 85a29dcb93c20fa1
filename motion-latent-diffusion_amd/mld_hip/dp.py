"""Data-parallel sampling across the GPUs of one node.

Every motion is independent end to end (attention never crosses samples, CFG pairs and DDIM are
per-sample), so the path shards over prompts with NO data-path collective: each rank holds a full weight
replica -- shipped once as ONE packed broadcast (RCCL over xGMI on MI355X; gloo in the CPU tests) -- and
samples its own contiguous block of prompts.  One process per GPU (torch.distributed).
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of n prompts for `rank`: sizes differ by at most one, order preserved."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_state(tensors: Dict[str, np.ndarray]) -> Tuple[np.ndarray, List[Tuple[str, Tuple[int, ...], int]]]:
    """Flatten a name->array dict into one float32 blob + an index (name, shape, offset)."""
    index, off, parts = [], 0, []
    for k, v in tensors.items():
        a = np.ascontiguousarray(v, np.float32)
        index.append((k, tuple(a.shape), off))
        parts.append(a.ravel())
        off += a.size
    return (np.concatenate(parts) if parts else np.zeros(0, np.float32)), index


def broadcast_state(tensors: Dict[str, np.ndarray], index_template: Dict[str, np.ndarray], device, src: int = 0,
                    group=None) -> Dict[str, torch.Tensor]:
    """ONE broadcast of the packed weights from `src`.  `tensors` is only read on `src`; every rank passes
    `index_template` (names + shapes, values ignored) so the blob layout is known without a second message."""
    import torch.distributed as dist
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    _, index = pack_state(index_template)
    total = sum(int(np.prod(s)) for _, s, _ in index)
    if rank == src:
        blob_np, _ = pack_state({k: tensors[k] for k, _, _ in index})
        blob = torch.from_numpy(blob_np).to(device)
    else:
        blob = torch.empty(total, dtype=torch.float32, device=device)
    if dist.is_initialized():        # also at world size 1: the collective path is the same code at every N
        dist.broadcast(blob, src=src, group=group)
    return {k: blob[o:o + int(np.prod(s))].view(*s) for k, s, o in index}


def gather_lengths_order(n: int, world: int) -> List[Tuple[int, int]]:
    return [shard_range(n, r, world) for r in range(world)]


class DataParallelSampler:
    """Shard a list of prompts over the ranks of the default process group; each rank runs the model on its block in
    chunks of `batch_size` and returns (global_indices, joints_list) for its block.

    in_flight > 1 (text-to-motion models on the fused engine path): consecutive chunks are issued on `in_flight` rotating
    HIP streams, so several batches overlap on the GPU (configure the engine with ``mld_hip.engine.configure("text", max_in_flight=in_flight)`` before
    its first use; with fewer workspaces the engine orders the calls behind each other on the device -- a workspace is
    never shared by two calls at once).  Results are identical either way.

    coalesce > 1: `coalesce` consecutive chunks go into ONE engine call (``MLD.sample_many`` -> ``mldhip_sample_many``: one chain over
    coalesce x batch_size motions on the throughput kernels); combined with in_flight = 4 this is the serving shape bench.py's
    headline measures (configure ``max_batch >= coalesce * batch_size``).  Set GPU_MAX_HW_QUEUES=8 before HIP initialises."""

    def __init__(self, model, batch_size: int = 64, in_flight: int = 1, coalesce: int = 1):
        self.model = model
        self.batch_size = batch_size
        self.in_flight = max(1, int(in_flight))
        self.coalesce = max(1, int(coalesce))

    def __call__(self, texts: Sequence[str], lengths: Sequence[int]):
        import torch.distributed as dist
        rank = dist.get_rank() if dist.is_initialized() else 0
        world = dist.get_world_size() if dist.is_initialized() else 1
        lo, hi = shard_range(len(texts), rank, world)
        chunks = [(s, min(hi, s + self.batch_size)) for s in range(lo, hi, self.batch_size)]
        m = self.model
        overlap = ((self.in_flight > 1 or self.coalesce > 1) and torch.cuda.is_available() and getattr(m, "fused", False)
                   and getattr(m, "condition", None) == "text" and getattr(m, "vae_type", "") != "no")   # plain text-to-motion only
        out = []
        if not overlap:
            for s, e in chunks:
                out.extend(m({"text": list(texts[s:e]), "length": list(lengths[s:e])}))
            return list(range(lo, hi)), out
        streams = [torch.cuda.Stream() for _ in range(self.in_flight)]
        for st in streams:
            st.wait_stream(torch.cuda.current_stream())
        pending = []
        groups = [chunks[g:g + self.coalesce] for g in range(0, len(chunks), self.coalesce)]
        for i, grp in enumerate(groups):
            with torch.cuda.stream(streams[i % self.in_flight]):
                reqs = []
                for s, e in grp:
                    tx, ln = list(texts[s:e]), [int(x) for x in lengths[s:e]]
                    reqs.append((m.text_encoder([""] * len(tx) + tx), ln))      # mld.py:224-231: unconditional half first
                if len(reqs) == 1:
                    joints, _, _ = m.sample(*reqs[0])
                    pending.append((joints, reqs[0][1]))
                else:
                    for (joints, _, _), (_, ln) in zip(m.sample_many(reqs), reqs):
                        pending.append((joints, ln))
        for st in streams:
            torch.cuda.current_stream().wait_stream(st)
        for joints, ln in pending:
            j = joints.cpu()
            out.extend(j[k, :n] for k, n in enumerate(ln))
        return list(range(lo, hi)), out
