"""Multi-GPU layer of the projection path: one process per GPU, independent face pairs sharded across ranks, one
collective at the end.

The reference has no multi-GPU projection at all: every driver pins one device (`CUDA_VISIBLE_DEVICES`,
1024_example_wing_loss_perceptual_sqz_MSE.py:213) and walks its image list serially
(projection_example_v2_percept_morph.py:329-365).  Each target is an independent search with private latent, noise stream
and best-so-far state, and the generator/LPIPS weights are read-only replicas, so the path shards with NO data-path
collective; the only exchange is the result gather {latent [k*D] f32, best loss, best step} per item (2.2 KB), done with a
single all_gather over RCCL/xGMI (backend "nccl" on ROCm) or gloo in CPU tests.
"""
from __future__ import annotations

import os

import torch

_STORE = None          # the key-value store the work queues count in: handed in by whoever created the process group (set_store / init_process_group)


def make_store(rank: int = None, world_size: int = None, host: str = None, port: int = None, timeout_s: float = 1800.0):
    """A `torch.distributed.TCPStore` on the job's rendezvous address (MASTER_ADDR / MASTER_PORT, or host / port): rank 0 serves it, the
    others connect -- or, under `torch.distributed.run`, whose agent already serves a store on that port (TORCHELASTIC_USE_AGENT_STORE),
    every rank connects to the agent's.  This is the store `init_process_group` below hands to torch AND keeps for the work queues."""
    import datetime
    import torch.distributed as dist
    rank = int(os.environ["RANK"]) if rank is None else int(rank)
    world_size = int(os.environ["WORLD_SIZE"]) if world_size is None else int(world_size)
    host = os.environ.get("MASTER_ADDR", "127.0.0.1") if host is None else host
    port = int(os.environ["MASTER_PORT"]) if port is None else int(port)
    agent = os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "") == "True"
    return dist.TCPStore(host, port, world_size, is_master=(rank == 0 and not agent), timeout=datetime.timedelta(seconds=timeout_s))


def set_store(store):
    """Hand the module the store shared by all ranks of the job (any torch.distributed.Store: the one passed to
    `torch.distributed.init_process_group(store=...)`, a TCPStore / FileStore of the caller's own; None forgets it).  The queues count
    under a prefix of their own that carries the launcher's restart count: the keys never collide with the rendezvous keys the same store
    may hold (under torch.distributed.run it is the agent's store), and after an elastic restart (max_restarts > 0) the new attempt does
    not see the counters the failed one left behind -- torch's own env rendezvous separates attempts the same way."""
    global _STORE
    if store is None:
        _STORE = None
        return None
    import torch.distributed as dist
    _STORE = dist.PrefixStore(f"mgf/attempt_{os.environ.get('TORCHELASTIC_RESTART_COUNT', '0')}", store)
    return store


def get_store():
    return _STORE


def init_process_group(backend: str, rank: int = None, world_size: int = None, host: str = None, port: int = None, store=None, **kw):
    """`torch.distributed.init_process_group(backend, store=..., rank=..., world_size=...)` on a store this module keeps a handle to
    (`make_store` unless one is passed in), so that `WorkQueue` never has to reach into torch's private state for it.  The drivers' entry
    points (bench.py, cli.py) create their process group through this call; a caller with a group of its own uses `set_store`."""
    import torch.distributed as dist
    rank = int(os.environ["RANK"]) if rank is None else int(rank)
    world_size = int(os.environ["WORLD_SIZE"]) if world_size is None else int(world_size)
    if store is None:
        store = make_store(rank, world_size, host, port)
    dist.init_process_group(backend, store=store, rank=rank, world_size=world_size, **kw)
    return set_store(store)


def shard_items(n_items: int, rank: int, world: int):
    """Static round-robin partition `items[rank::world]`."""
    return list(range(rank, n_items, world))


class WorkQueue:
    """Dynamic partition for ragged workloads (targets whose steps are skipped when no face is found, early exits): every rank
    pulls the next item index from one shared atomic counter in the process group's key-value store.  No
    tensor collective is involved; with no process group it degenerates to range(n_items).

    Every queue has its own counter key: `name` plus an epoch that counts the queues of that name created in this process.  Ranks
    create their queues in the same order (project_many is called collectively), so the epoch -- and the key -- agree across ranks,
    and a second queue in the same process group (the second stage of a two-stage run) starts from zero again."""

    _epochs: dict = {}

    def __init__(self, n_items: int, name: str = "mgf_queue", group=None, store=None):
        """group: the process group whose ranks share this queue (default: all ranks).  The counter lives in the job's rendezvous
        store whatever the group; a sub-group's key carries its global ranks, so disjoint sub-groups draw from separate counters and
        the queue's population is exactly the group `run_sharded` gathers over.
        store: the key-value store shared by the ranks (default: the one this module was handed, `set_store` / `init_process_group`)."""
        import torch.distributed as dist
        on = dist.is_available() and dist.is_initialized()
        if on and group is not None:
            name = f"{name}@{'-'.join(str(r) for r in dist.get_process_group_ranks(group))}"
        epoch = WorkQueue._epochs.get(name, 0)
        WorkQueue._epochs[name] = epoch + 1
        self.n_items, self.key = n_items, f"{name}/{epoch}/next"
        self.store = None
        self._local = 0
        if on and dist.get_world_size(group) > 1:
            self.store = store if store is not None else _STORE
            if self.store is None:
                raise RuntimeError("WorkQueue: more than one rank but no shared store: create the process group with "
                                   "morphganformer_amd.distributed.init_process_group(...), or hand the store over with set_store(store) / "
                                   "WorkQueue(..., store=store)")

    def __iter__(self):
        return self

    def __next__(self) -> int:
        if self.store is None:
            i = self._local
            self._local += 1
        else:
            i = self.store.add(self.key, 1) - 1            # atomic fetch-and-add on the rendezvous store
        if i >= self.n_items:
            raise StopIteration
        return i


def pack_result(latent: torch.Tensor, best_loss: float, best_step: int, item: int = 0) -> torch.Tensor:
    """[k*D + 3] float64 record: latent (exact: f32 embeds in f64), loss, step, item id."""
    flat = latent.reshape(-1).double()
    tail = torch.tensor([best_loss, float(best_step), float(item)], dtype=torch.float64, device=flat.device)
    return torch.cat([flat, tail])


def unpack_results(records: torch.Tensor, latent_shape):
    n = records.shape[0]
    k = 1
    for s in latent_shape:
        k *= s
    return {"latents": records[:, :k].float().reshape(n, *latent_shape), "losses": records[:, k].clone(),
            "steps": records[:, k + 1].long(), "items": records[:, k + 2].long()}


def gather_results(latent: torch.Tensor, best_loss: float, best_step: int, item: int = 0, group=None):
    """all_gather of one record per rank; returns the unpacked dict on every rank (works without an initialised
    process group for world size 1)."""
    import torch.distributed as dist
    rec = pack_result(latent, best_loss, best_step, item)
    if not (dist.is_available() and dist.is_initialized()):
        return unpack_results(rec[None], tuple(latent.shape[-2:]))
    world = dist.get_world_size(group)
    out = torch.empty(world * rec.numel(), dtype=rec.dtype, device=rec.device)      # flat: gloo insists on 1-D in/out
    dist.all_gather_into_tensor(out, rec.contiguous(), group=group)
    return unpack_results(out.view(world, rec.numel()), tuple(latent.shape[-2:]))


def gather_many(records: torch.Tensor, counts_max: int, group=None):
    """Ragged gather for several items per rank: records [m, R] padded to counts_max rows with item id -1."""
    import torch.distributed as dist
    m, r = records.shape
    pad = torch.full([counts_max, r], -1.0, dtype=records.dtype, device=records.device)
    pad[:m] = records
    if not (dist.is_available() and dist.is_initialized()):
        return pad[:m]
    world = dist.get_world_size(group)
    out = torch.empty(world * counts_max * r, dtype=records.dtype, device=records.device)
    dist.all_gather_into_tensor(out, pad.reshape(-1), group=group)
    out = out.view(world * counts_max, r)
    keep = out[:, -1] >= 0
    out = out[keep]
    return out[torch.argsort(out[:, -1])]


def run_sharded(n_items: int, work_fn, record_width: int, device, dynamic: bool = False, queue_name: str = "mgf_queue", group=None, store=None):
    """The multi-GPU skeleton of `drivers.project_many`, free of any GPU work so that it runs under gloo on CPU: this rank takes its
    items -- `items[rank::world]`, or with dynamic=True whatever the shared WorkQueue hands it (ragged per-item cost: targets whose
    steps are skipped when no face is found) -- calls `work_fn(item) -> record [record_width] float64` (pack_result) for each, and ONE
    ragged all_gather returns every item's record to every rank, ordered by item id.  Returns (records [n_items, width], mine) where
    `mine` lists the items this rank worked on, in the order it did."""
    import torch.distributed as dist
    on = dist.is_available() and dist.is_initialized()
    rank, world = (dist.get_rank(group), dist.get_world_size(group)) if on else (0, 1)
    order = WorkQueue(n_items, name=queue_name, group=group, store=store) if dynamic else shard_items(n_items, rank, world)
    recs, mine = [], []
    for i in order:
        recs.append(work_fn(i))
        mine.append(i)
    rows = torch.stack(recs) if recs else torch.empty([0, record_width], dtype=torch.float64, device=device)
    assert rows.shape[1] == record_width, (tuple(rows.shape), record_width)
    return gather_many(rows, n_items if dynamic else -(-n_items // world), group=group), mine
