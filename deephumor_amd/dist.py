"""Image-sharded captioning over the GPUs of one node (one process per GPU, RCCL over xGMI).

Every image's caption depends only on that image and the replicated weights (the reference's
``generate`` is batch-1, so independence holds by construction -- SURVEY.md section 8e).  The image
batch is therefore split into contiguous shards, each rank decodes its shard with NO data-path
collective, and the results are exchanged ONCE per batch with a single ``all_gather`` of the padded
token ids and lengths (64 KB per rank at 256 x 32) -- never per decode step.  The Philox noise of
stochastic beam search is keyed by the GLOBAL image index (``img0``), so captions do not depend on
the world size.
"""
import os

import torch
import torch.distributed as dist

__all__ = ["shard_range", "gather_captions", "gather_captions_async", "generate_sharded", "generate_micro_sharded"]


def shard_range(n_total, rank, world_size):
    """Contiguous shard ``[lo, hi)`` of ``n_total`` images for ``rank``; the first ``n_total %
    world_size`` ranks get one extra image (300 images / 8 ranks -> 38,38,38,38,37,37,37,37)."""
    base, extra = divmod(n_total, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _skip_collective(group, always):
    """A single rank has nothing to exchange: the collective is skipped unless ``always`` (or ``DH_DIST_ALWAYS=1``) asks for it --
    which is how the RCCL path is exercised on a one-GPU box (``bench.py --rccl-single``, ``tests/test_dist_gpu.py``)."""
    if not (dist.is_available() and dist.is_initialized()):
        return True
    if always is None:
        try:
            from . import hip
            always = bool(hip.option("dist_always"))
        except (RuntimeError, OSError):    # gloo / CPU ranks on a box without the GPU library: the option's environment default
            always = os.environ.get("DH_DIST_ALWAYS", "0") not in ("", "0")
    return dist.get_world_size(group) == 1 and not always


class PendingGather:
    """Handle of ``gather_captions_async``: ``wait()`` -> ``(tokens [n_total, T], lengths [n_total])``.  With RCCL ``wait`` makes the
    CURRENT STREAM wait for the collective (the host does not block), so a caller that issues batch i + 1's decode before waiting
    for batch i's exchange hides the exchange behind that decode."""

    def __init__(self, work, out, n_total, world, cap, t, direct=None):
        self.work, self.out, self.n_total, self.world, self.cap, self.t, self.direct = work, out, n_total, world, cap, t, direct

    def wait(self):
        if self.direct is not None:
            return self.direct
        if self.work is not None:
            self.work.wait()
            self.work = None
        out, t = self.out, self.t
        if self.n_total == self.world * self.cap:           # even shards: the gathered buffer IS the global order
            full = out
        else:
            rows = []
            for r in range(self.world):
                lo, hi = shard_range(self.n_total, r, self.world)
                rows.append(out[r * self.cap:r * self.cap + (hi - lo)])
            full = torch.cat(rows, 0)
        self.direct = (full[:, :t].contiguous(), full[:, t].contiguous())
        return self.direct


def gather_captions_async(tokens, lengths, n_total, group=None, always=None):
    """``gather_captions`` with the collective issued asynchronously (``async_op=True``): returns a ``PendingGather``."""
    if _skip_collective(group, always):
        return PendingGather(None, None, n_total, 1, n_total, tokens.shape[1], direct=(tokens, lengths))
    world = dist.get_world_size(group)
    t = tokens.shape[1]
    cap = -(-n_total // world)
    if tokens.shape[0] == cap:
        packed = torch.empty((cap, t + 1), dtype=torch.int64, device=tokens.device)
    else:
        packed = torch.zeros((cap, t + 1), dtype=torch.int64, device=tokens.device)
    packed[:tokens.shape[0], :t] = tokens
    packed[:tokens.shape[0], t] = lengths
    out = torch.empty((world * cap, t + 1), dtype=torch.int64, device=tokens.device)
    work = dist.all_gather_into_tensor(out, packed, group=group, async_op=True)
    return PendingGather(work, out, n_total, world, cap, t)


def gather_captions(tokens, lengths, n_total, group=None, always=None):
    """``tokens [n_local, T]`` int64, ``lengths [n_local]`` -> the full ``[n_total, T]`` / ``[n_total]``
    on every rank, in global image order.  One ``all_gather_into_tensor`` on shards padded to the
    largest shard (uneven shards differ by at most one image)."""
    return gather_captions_async(tokens, lengths, n_total, group, always).wait()


def generate_sharded(generate_fn, n_total, group=None, always=None):
    """Runs ``generate_fn(lo, hi) -> (tokens, lengths)`` on this rank's shard (``lo`` is the global
    index of its first image: pass it as ``img0``) and gathers the whole batch on every rank."""
    rank = dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    lo, hi = shard_range(n_total, rank, world)
    tokens, lengths = generate_fn(lo, hi)
    return gather_captions(tokens, lengths, n_total, group, always)


def generate_micro_sharded(generate_fn, n_total, n_shards, group=None, always=None):
    """A global batch cut into MORE shards than there are ranks (BASELINE config C4's 2,048 images as 8 shards of 256 on fewer than
    8 GPUs -- on one GPU: 8 shards one after another): shard ``s`` = ``shard_range(n_total, s, n_shards)``, rank ``r`` of ``W``
    decodes shards ``r * (n_shards / W) ...`` in order, and after each of its shards the ranks exchange that round's shards with
    the same single ``all_gather`` as ``gather_captions``.  Every image keeps its GLOBAL index (``generate_fn(lo, hi)`` must pass
    ``img0=lo``), so the captions equal those of an ``n_shards``-rank run and of one big batch.  Returns the whole
    ``(tokens [n_total, T], lengths [n_total])`` on every rank."""
    rank = dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    if n_shards % world:
        raise ValueError(f"n_shards ({n_shards}) must be a multiple of the world size ({world})")
    per = n_shards // world
    spans = [shard_range(n_total, s, n_shards) for s in range(n_shards)]
    got = [None] * n_shards
    for j in range(per):
        lo, hi = spans[rank * per + j]
        toks, lens = generate_fn(lo, hi)
        # this round holds shards {r * per + j}: a sub-batch whose shards are NOT contiguous in the global order, so it is gathered
        # as raw padded payloads (the largest shard of the whole batch sets the padding) and re-assembled below
        t = toks.shape[1]
        cap = max(b - a for a, b in spans)
        packed = torch.zeros((cap, t + 1), dtype=torch.int64, device=toks.device)
        packed[:toks.shape[0], :t] = toks
        packed[:toks.shape[0], t] = lens
        if _skip_collective(group, always):
            out = packed
        else:
            out = torch.empty((world * cap, t + 1), dtype=torch.int64, device=toks.device)
            dist.all_gather_into_tensor(out, packed, group=group)
        for r in range(world):
            a, b = spans[r * per + j]
            got[r * per + j] = out[r * cap:r * cap + (b - a)]
    full = torch.cat(got, 0)
    return full[:, :-1].contiguous(), full[:, -1].contiguous()


def gather_rows(x, n_total, group=None, always=None):
    """``x [n_local, ...]`` (any dtype: logits, log-probabilities, perplexities) -> ``[n_total, ...]`` on every
    rank in global row order: the north star's "all-gather of logits" -- one ``all_gather_into_tensor`` per call on
    shards padded to the largest one (contiguous ``shard_range`` shards)."""
    if _skip_collective(group, always):
        return x
    world = dist.get_world_size(group)
    cap = -(-n_total // world)
    packed = torch.zeros((cap,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    packed[:x.shape[0]] = x
    out = torch.empty((world * cap,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out, packed.contiguous(), group=group)
    rows = []
    for r in range(world):
        lo, hi = shard_range(n_total, r, world)
        rows.append(out[r * cap:r * cap + (hi - lo)])
    return torch.cat(rows, 0)


def score_sharded(score_fn, n_total, group=None, always=None):
    """Teacher-forced scoring sharded by caption batch: ``score_fn(lo, hi) -> [hi - lo, ...]`` (e.g. per-caption
    perplexities from ``experiments.scoring.score_captions`` on captions ``lo..hi``) on this rank's contiguous
    shard, then one all-gather so that every rank holds all ``n_total`` rows."""
    rank = dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    lo, hi = shard_range(n_total, rank, world)
    return gather_rows(score_fn(lo, hi), n_total, group, always)
