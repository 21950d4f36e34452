"""Multi-GPU driver: independent images shard over ranks; the only exchange of the eval path is one all-gather
of per-image (bpp, psnr, mse, bits) rows at the end (RCCL over xGMI when the backend is "nccl").

The reference has no distribution at all (one GPU per Slurm array task, slurm_template.py:8-11);
images are independent (mshyper/models.py:425-433), so there is no data-path collective.

The training step is data-parallel: every rank holds a full replica, takes its own slice of the
batch, and the gradients are averaged with a handful of large all-reduces (``BucketReducer``) that are launched
while the backward pass is still producing the earlier layers' gradients.
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.distributed as dist


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment; (0, 0, 1) when not launched by it."""
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)))


def init(backend=None):
    """Initialise torch.distributed if WORLD_SIZE > 1 (nccl == RCCL on ROCm; gloo for CPU tests).  A backend named
    explicitly -- the argument or SNTC_DIST_BACKEND -- also forms a ONE-rank group, so that the collectives of the path run
    through RCCL on a single GPU (the only RCCL evidence obtainable on a one-GPU box; bench.py does this by default)."""
    rank, local_rank, world = env_world()
    explicit = backend or os.environ.get("SNTC_DIST_BACKEND")
    if (world > 1 or explicit) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            backend = os.environ.get("SNTC_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shutdown():
    if dist.is_initialized():
        dist.destroy_process_group()


def describe_world(device=None):
    """Evidence that the collective backend saw every rank: {backend, world, rccl_version, devices: one entry per
    rank (index, name, gcnArchName, PCI bus id), collected with ONE all_gather_object}.  ``device`` None = CPU process."""
    rank, local_rank, world = env_world()
    me = dict(rank=rank, host_pid=os.getpid())
    if device is not None and torch.cuda.is_available():
        props = torch.cuda.get_device_properties(device)
        me.update(device=torch.cuda.current_device(), name=props.name, arch=getattr(props, "gcnArchName", None),
                  pci_bus_id=getattr(props, "pci_bus_id", None), cus=props.multi_processor_count)
    else:
        me.update(device="cpu")
    backend, version = None, None
    devices = [me]
    if dist.is_initialized():
        backend = dist.get_backend()
        devices = [None] * dist.get_world_size()
        dist.all_gather_object(devices, me)
        if backend == "nccl":
            try:
                version = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:
                version = None
    return dict(backend=backend if backend != "nccl" else "nccl (RCCL on ROCm)", world=world, rccl_version=version,
                devices=devices)


def deal_units(num_units, rank, world):
    """Independent work units (images, (lambda, image) pairs, SGA batches) -> the ids this rank owns: unit u goes to
    rank u mod world.  No unit is dropped or duplicated for any (num_units, world)."""
    return list(range(rank, num_units, world))


def shard_indices(num_items, rank, world):
    """Image i goes to rank i mod world (round-robin keeps the Kodak orientations balanced)."""
    return list(range(rank, num_items, world))


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device="cpu"):
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_ranks(value, device="cpu"):
    """Every rank's ``value`` as a list in rank order, on every rank (one all_gather of a float64): the per-rank step times an
    N-GPU bench line carries so that a skewed rank is visible next to the maximum."""
    if not dist.is_initialized():
        return [float(value)]
    dev = "cpu" if dist.get_backend() == "gloo" else device
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def share_from_rank0(obj):
    """Rank 0's ``obj`` on every rank (one ``all_gather_object``; the payloads here are a few KB of schedule choices)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return obj
    got = [None] * dist.get_world_size()
    dist.all_gather_object(got, obj if dist.get_rank() == 0 else None)
    return got[0]


def gather_rows(local_rows, indices, num_items, device="cpu", width=None):
    """All-gather per-image metric rows.  local_rows: float array [len(indices), k] for the image
    ids in ``indices``.  Returns the full [num_items, k] table on every rank, ordered by image id.
    ``width``: k, for a rank that may hold no rows at all."""
    local_rows = np.asarray(local_rows, np.float64)
    local_rows = local_rows.reshape(len(indices), -1) if local_rows.size else np.zeros((0, width or 0))
    k = local_rows.shape[1]
    if not dist.is_initialized():
        out = np.full((num_items, k), np.nan)
        out[np.asarray(indices, int)] = local_rows
        return out
    world = dist.get_world_size()
    per = -(-num_items // world)                                    # pad every rank to the same length
    buf = torch.full((per, k + 1), float("nan"), dtype=torch.float64, device=device)
    if len(indices):
        buf[:len(indices), 0] = torch.as_tensor(indices, dtype=torch.float64)
        buf[:len(indices), 1:] = torch.as_tensor(local_rows)
    if dist.get_backend() == "gloo":
        buf = buf.cpu()
    gathered = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(gathered, buf)
    out = np.full((num_items, k), np.nan)
    for g in gathered:
        g = g.cpu().numpy()
        ok = ~np.isnan(g[:, 0])
        out[g[ok, 0].astype(int)] = g[ok, 1:]
    return out


def run_units(num_units, fn, device="cpu", width=None):
    """The sharding driver of every multi-GPU evaluation in this repository (Kodak R-D sweep = (lambda, image) units,
    BASELINE.json configs[3], reference launch.py:49-50 + mshyper/configs/jpegl.py get_hyper; SGA on Tecnick in batches
    of 5 = one unit per batch, configs[4], reference common/itinf_lib.py:187-207): unit u -> rank u mod world
    (``deal_units``), ``fn(u)`` -> a row of floats on the owning rank, then ONE all-gather of the rows.  Returns the
    full [num_units, width] table on every rank, ordered by unit id -- identical for any world size, because a unit's row
    depends on the unit alone."""
    rank, _, world = env_world()
    if not dist.is_initialized():
        rank, world = 0, 1
    mine = deal_units(num_units, rank, world)
    rows = [np.asarray(fn(u), np.float64).ravel() for u in mine]
    if width is None:
        width = len(rows[0]) if rows else 0
        if dist.is_initialized():                       # a rank without units must still learn the row width
            w = torch.tensor([width], dtype=torch.int64, device="cpu" if dist.get_backend() == "gloo" else device)
            dist.all_reduce(w, op=dist.ReduceOp.MAX)
            width = int(w.item())
    local = np.stack(rows) if rows else np.zeros((0, width))
    return gather_rows(local, mine, num_units, device=device, width=width)


class BucketReducer:
    """Bucketed, overlapped all-reduce of ONE flat gradient buffer.

    ``slices``: ordered {bucket name: (lo, hi)} over the flat buffer, in the order the backward pass completes them.
    ``launch(name)`` starts the (asynchronous) sum of that slice across ranks as soon as its last gradient is written
    -- on RCCL the collective runs on its own stream, so it overlaps the rest of the backward pass; xGMI is point to
    point, so a few large buckets (tens of MB) keep every ring link busy instead of paying the latency per tensor.
    ``finish()`` waits for all of them and returns the factor (1 / world) that turns the sums into means -- the caller
    folds it into its optimizer kernel instead of spending another pass over the buffer."""

    def __init__(self, flat, slices):
        self.flat, self.slices = flat, dict(slices)
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.active = dist.is_initialized()        # a one-rank group still runs its all-reduces through the backend
        self._handles = []
        self._launched = []

    def launch(self, name):
        lo, hi = self.slices[name]
        self._launched.append(name)
        if self.active and hi > lo:
            self._handles.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, async_op=True))

    def finish(self):
        missing = [n for n in self.slices if n not in self._launched]
        if missing:
            raise RuntimeError(f"gradient buckets never launched: {missing}")
        for h in self._handles:
            h.wait()
        self._handles, self._launched = [], []
        return 1.0 / self.world
