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


def pack_range(n: int, rank: int, world: int, per_rank: int) -> Tuple[int, int]:
    """"pack" sharding: ceil(n / per_rank) BUSY ranks take `per_rank` prompts each (the last one the remainder), the other ranks none -- fewer, bigger engine calls
    (one coalesced call of a few hundred motions runs the persistent loop at a higher rate per GPU than bs-64 calls do on the launch family)."""
    per_rank = max(1, int(per_rank))
    lo = min(n, rank * per_rank)
    return lo, min(n, lo + per_rank)


def plan_shards(n: int, world: int, batch_size: int, max_batch: int, policy: str = "auto", single_batch_is_fast: bool = True) -> dict:
    """How DataParallelSampler spreads n prompts over `world` ranks.  "spread": contiguous equal shards on every rank (the reference's own multi-GPU form,
    scripts/fit_motion_parallel.sh; BASELINE config 3 = one bs-64 batch per rank).  "pack": as few ranks as hold the prompts at `max_batch` per rank.  "auto":
    spread whenever a bs-64 call is served by the cluster loop (engine >= r05: 8.1 k motions/s per rank at bs 64, 8 ranks = 65 k, against 20.5 k for one rank
    with all 512 prompts in one call) or the shards are not smaller than a coalesced call anyway; pack only when per-rank shards would be single small batches
    on an engine without that path.  Returns {"policy", "busy_ranks", "prompts_per_busy_rank", "why"}."""
    per_spread = -(-n // max(1, world))
    if policy == "auto":
        small = per_spread <= batch_size and world > 1
        policy = "pack" if (small and not single_batch_is_fast and max_batch > batch_size) else "spread"
        why = ("auto: shards of %d prompts per rank; a single batch %s the cluster loop" % (per_spread, "runs" if single_batch_is_fast else "does not run"))
    else:
        why = "forced"
    if policy == "pack":
        per = min(max(batch_size, max_batch), n)
        if -(-n // per) > world:
            # more prompts than `world` calls of max_batch hold: every rank is busy and takes ceil(n / world) (its shard is chunked by batch_size and coalesced
            # per engine call anyway) -- the packed ranges must cover EVERY prompt (advisor r5: n = 512, world = 1, max_batch = 64 used to cover 64)
            per = -(-n // world)
            why += "; more prompts than world x max_batch: ceil(n / world) per rank"
        return {"policy": "pack", "busy_ranks": min(world, -(-n // per)), "prompts_per_busy_rank": per, "why": why}
    return {"policy": "spread", "busy_ranks": min(world, n), "prompts_per_busy_rank": per_spread, "why": why}


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
    """Shard a list of prompts (or action labels) over the ranks of the default process group; each rank runs the model on its
    block in chunks of `batch_size` and returns (global_indices, motions) for its block -- joints [len, 22, 3] per prompt for the
    text models (``MLD.forward``, mld.py:216-265; also the diffusion-only variant), features [len, nfeats] per label for the
    action model (``MLD.a2m_eval``, mld.py:710-735: its joints need SMPL).

    `init_latents` (optional, indexed like the prompts: [N, 1, D] latents, or [N, Tmax, nfeats] raw-motion noise for the
    diffusion-only variant) pins the starting noise per PROMPT, so a motion does not depend on the world size or on which rank /
    chunk it lands in -- what the multi-process tests compare.  Without it every chunk draws from torch's generator.
    (Diffusion-only variant: the reference's trans_dec denoiser attends over the padded batch, so there a motion also depends on the
    Tmax of the chunk it is sampled in -- reference behaviour, reproduced; equal-Tmax chunks give shard-invariant motions.)

    in_flight > 1 (text-to-motion models on the fused engine path, latent and diffusion-only): consecutive chunks are issued on `in_flight` rotating
    HIP streams, so several batches overlap on the GPU (configure the engine with ``mld_hip.engine.configure("text" / "novae", max_in_flight=in_flight)`` before
    its first use; with fewer workspaces the engine orders the calls behind each other on the device -- a workspace is
    never shared by two calls at once).  Results are identical either way.  The diffusion-only variant gains most (its big GEMM tiles and its
    attention run one workgroup per CU and leave gaps a second batch fills: 5.0 instead of 5.8 ms per DDPM step and batch on MI355X with
    in_flight = 2; more does not help) -- provided ROCm put the two streams on different hardware queues (it maps HIP streams onto 4 queues;
    bench.py shows the one-line probe).

    coalesce: how many consecutive chunks go into ONE engine call (``MLD.sample_many`` -> ``mldhip_sample_many``: one chain over
    coalesce x batch_size motions; from 192 motions per call the split-f16 engine runs the reverse loop as one persistent launch, a
    workgroup per 8 motions, whose run time does not depend on the batch up to 2 048 motions).  Default 1: one chunk per engine call,
    the reference's own shape (``MLD.forward`` per batch) -- coalesced calls run other loop kernels and are tolerance-equal, not
    bit-equal, to per-chunk calls, so the switch is the caller's (advisor r4).  ``"auto"`` (``None`` is accepted as a synonym):
    as many chunks of this rank's shard as the engine's capacity holds, ``min(chunks, engine.max_batch // batch_size)`` -- BASELINE
    config 3 (512 prompts) on ONE rank is one 512-motion call instead of eight latency-kernel calls when the engine was configured
    with ``max_batch >= 512``; on eight ranks every rank holds one bs-64 batch and there is nothing to coalesce (the literal
    ``MLD.forward`` shape).  An int forces that many (the engine refuses more motions than its ``max_batch``)."""

    def __init__(self, model, batch_size: int = 64, in_flight: int = 1, coalesce=1, shard: str = "auto", verbose: bool = False):
        self.shard, self.verbose, self.last_plan = shard, verbose, None     # shard: "auto" | "spread" | "pack" (plan_shards)
        self.model = model
        self.batch_size = batch_size
        self.in_flight = max(1, int(in_flight))
        self.coalesce = None if (coalesce is None or coalesce == "auto") else max(1, int(coalesce))

    def pick_coalesce(self, nchunks: int) -> int:
        """The automatic rule: chunks per engine call from the shard size and the engine's capacity."""
        if self.coalesce is not None and not (self.last_plan and self.last_plan["policy"] == "pack"):      # packed shards exist to be coalesced
            return self.coalesce
        try:
            cap = int(self.model._engine().cfg.max_batch) // self.batch_size
        except Exception:          # not a fused Hip* model (no engine behind it): one chunk per call, like the reference
            return 1
        return max(1, min(nchunks, cap))

    def __call__(self, texts: Sequence[str] = None, lengths: Sequence[int] = None, actions: Sequence[int] = None, init_latents=None,
                 step_noise=None):
        """step_noise (diffusion-only variant, optional): the DDPM scheduler's per-step draws [steps, N, Tmax, nfeats], indexed
        like the prompts -- with `init_latents` it makes a motion independent of how the prompts are sharded and chunked."""
        import torch.distributed as dist
        m = self.model
        action = getattr(m, "condition", None) == "action"
        items = actions if action else texts
        if items is None or lengths is None or len(items) != len(lengths):
            raise ValueError("DataParallelSampler needs %s and lengths of equal size" % ("actions" if action else "texts"))
        rank = dist.get_rank() if dist.is_initialized() else 0
        world = dist.get_world_size() if dist.is_initialized() else 1
        try:
            eng = m._engine()
            max_batch, prec = int(eng.cfg.max_batch), int(eng.cfg.precision)
        except Exception:          # not a fused Hip* model
            max_batch, prec = self.batch_size, 0
        fast_single = prec == 1 and self.batch_size <= 128 and getattr(m, "vae_type", "") != "no"      # F16X3 engine: calls of <= 128 motions run the cluster loop
        self.last_plan = plan_shards(len(items), world, self.batch_size, max_batch, self.shard, fast_single)
        if self.verbose and rank == 0:
            print("DataParallelSampler: %(policy)s over %(busy_ranks)d busy rank(s), %(prompts_per_busy_rank)d prompts each (%(why)s)" % self.last_plan, flush=True)
        if self.last_plan["policy"] == "pack":
            lo, hi = pack_range(len(items), rank, world, self.last_plan["prompts_per_busy_rank"])
        else:
            lo, hi = shard_range(len(items), rank, world)
        chunks = [(s, min(hi, s + self.batch_size)) for s in range(lo, hi, self.batch_size)]
        dev = next(m.parameters()).device
        novae = getattr(m, "vae_type", "") == "no"

        def noise(s, e, ln):
            if init_latents is None:
                return None
            z = init_latents[s:e]
            if novae:
                z = z[:, :max(ln)]                   # the chunk's own Tmax (mld.py:296-301)
            return z.to(dev).float().contiguous()

        can_overlap = (torch.cuda.is_available() and getattr(m, "fused", False)
                       and getattr(m, "condition", None) == "text" and not novae)         # latent text-to-motion only
        can_coalesce_action = torch.cuda.is_available() and getattr(m, "fused", False) and action and hasattr(m, "sample_many_action")
        coalesce = self.pick_coalesce(len(chunks)) if (can_overlap or can_coalesce_action) else 1
        self.last_coalesce = coalesce
        overlap = can_overlap and (self.in_flight > 1 or coalesce > 1)
        overlap_novae = (torch.cuda.is_available() and getattr(m, "fused", False) and novae and self.in_flight > 1 and len(chunks) > 1
                         and getattr(m, "do_classifier_free_guidance", True) and getattr(m, "condition", None) == "text")
        out = []
        if can_coalesce_action and coalesce > 1:
            # action model: `coalesce` chunks per engine call (MLD.sample_many_action -> mldhip_sample_many)
            for g0 in range(0, len(chunks), coalesce):
                grp = chunks[g0:g0 + coalesce]
                reqs = [([int(a) for a in actions[s:e]], [int(x) for x in lengths[s:e]]) for s, e in grp]
                lats = [noise(s, e, ln) for (s, e), (_, ln) in zip(grp, reqs)] if init_latents is not None else None
                for (feats, _), (_, ln) in zip(m.sample_many_action(reqs, init_latents=lats, device=dev), reqs):
                    f = feats.cpu()
                    out.extend(f[k, :n] for k, n in enumerate(ln))
            return list(range(lo, hi)), out
        if not overlap and not overlap_novae:
            for s, e in chunks:
                ln = [int(x) for x in lengths[s:e]]
                if action:
                    acts = [int(a) for a in actions[s:e]]
                    rs = m.a2m_eval({"action": torch.tensor(acts, device=dev).reshape(-1, 1), "length": ln}, init_latents=noise(s, e, ln))
                    feats = rs["m_rst"].cpu()
                    out.extend(feats[k, :n] for k, n in enumerate(ln))
                elif novae and step_noise is not None:
                    sn = step_noise[:, s:e, :max(ln)].to(dev).float().contiguous()
                    out.extend(m({"text": list(texts[s:e]), "length": ln}, init_latents=noise(s, e, ln), step_noise=sn))
                else:
                    out.extend(m({"text": list(texts[s:e]), "length": ln}, init_latents=noise(s, e, ln)))
            return list(range(lo, hi)), out
        streams = [torch.cuda.Stream() for _ in range(self.in_flight)]
        for st in streams:
            st.wait_stream(torch.cuda.current_stream())
        pending = []
        if overlap_novae:
            # diffusion-only variant: one mldhip_sample_novae call per chunk, chunks on rotating streams (each call's host work -- text encoder,
            # noise slices -- runs on its stream too); the results are fetched after the last call is enqueued
            for i, (s, e) in enumerate(chunks):
                with torch.cuda.stream(streams[i % self.in_flight]):
                    tx, ln = list(texts[s:e]), [int(x) for x in lengths[s:e]]
                    sn = step_noise[:, s:e, :max(ln)].to(dev).float().contiguous() if step_noise is not None else None
                    joints, _ = m.sample_novae(m.text_encoder([""] * len(tx) + tx), ln, noise(s, e, ln), sn)
                    pending.append((joints, ln))
            for st in streams:
                torch.cuda.current_stream().wait_stream(st)
            for joints, ln in pending:
                j = joints.cpu()
                out.extend(j[k, :n] for k, n in enumerate(ln))
            return list(range(lo, hi)), out
        groups = [chunks[g:g + coalesce] for g in range(0, len(chunks), coalesce)]
        for i, grp in enumerate(groups):
            with torch.cuda.stream(streams[i % self.in_flight]):
                reqs, lats = [], []
                for s, e in grp:
                    tx, ln = list(texts[s:e]), [int(x) for x in lengths[s:e]]
                    reqs.append((m.text_encoder([""] * len(tx) + tx), ln))      # mld.py:224-231: unconditional half first
                    lats.append(noise(s, e, ln))
                if len(reqs) == 1:
                    joints, _, _ = m.sample(*reqs[0], init_latents=lats[0])
                    pending.append((joints, reqs[0][1]))
                else:
                    for (joints, _, _), (_, ln) in zip(m.sample_many(reqs, init_latents=lats if init_latents is not None else None), reqs):
                        pending.append((joints, ln))
        for st in streams:
            torch.cuda.current_stream().wait_stream(st)
        for joints, ln in pending:
            j = joints.cpu()
            out.extend(j[k, :n] for k, n in enumerate(ln))
        return list(range(lo, hi)), out
