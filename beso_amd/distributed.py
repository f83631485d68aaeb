"""Data-parallel plumbing: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" in the CPU tests).

Inference shards the batch and needs no collective.  The training step has exactly one exchange:
the all-reduce of the score-matching gradients.  Gradients are reduced as ONE flat
fp32 bucket (9.4 M values = 37.5 MB for the kitchen model): on the fully connected xGMI mesh RCCL
splits a single large all-reduce over all seven links, which beats many per-tensor collectives that
are each latency bound.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def is_distributed() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def init_from_env(backend: Optional[str] = None) -> bool:
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (torchrun)."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    if ws <= 1 or (dist.is_available() and dist.is_initialized()):
        return ws > 1
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group(backend=backend, rank=int(os.environ["RANK"]), world_size=ws)
    return True


def shard_range(total: int, n_shards: int, index: int):
    """[lo, hi) of shard ``index`` when ``total`` independent samples are split over ``n_shards``
    ranks (remainder spread over the first ranks)."""
    base, rem = divmod(total, n_shards)
    lo = index * base + min(index, rem)
    return lo, lo + base + (1 if index < rem else 0)


class GradientBucket:
    """Flat fp32 gradient bucket for the C1 all-reduce.  ``sync(params)`` averages ``p.grad`` over the
    ranks in place; with ``async_op=True`` the caller overlaps the collective with other work and
    calls ``wait()`` before the optimizer step."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else torch.device("cpu")
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self._work = None

    def _pack(self):
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                self.flat[off:off + n].zero_()
            else:
                self.flat[off:off + n].copy_(p.grad.reshape(-1))
            off += n

    def _unpack(self):
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                p.grad = self.flat[off:off + n].view_as(p).clone()
            else:
                p.grad.copy_(self.flat[off:off + n].view_as(p))
            off += n

    def sync(self, async_op: bool = False, decomposed: bool = False):
        """``decomposed``: the same sum as reduce-scatter + all-gather (the two halves of the sharded exchange, for
        optimizers that step on all parameters)."""
        if not is_distributed():
            return None
        self._pack()
        self.flat.div_(world_size())            # mean over the GLOBAL batch (score_wrappers.py:79)
        if decomposed:
            n, w = self.flat.numel(), world_size()
            s = shard_size(n, w)
            padded = torch.zeros(s * w, dtype=self.flat.dtype, device=self.flat.device)
            padded[:n].copy_(self.flat)
            shard = torch.empty(s, dtype=self.flat.dtype, device=self.flat.device)
            reduce_scatter_flat(padded, shard)
            all_gather_flat(padded, shard)
            self.flat.copy_(padded[:n])
            self._unpack()
            return None
        self._work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=async_op)
        if not async_op:
            self._unpack()
        return self._work

    def wait(self):
        if self._work is not None:
            self._work.wait()
            self._work = None
            self._unpack()


def all_reduce_sum(flat: torch.Tensor) -> None:
    """C1 on an already averaged (pre-scaled by 1/world) flat gradient buffer: one collective, in place."""
    if is_distributed():
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)


def all_reduce_sum_overlapped(flat: torch.Tensor, early_range, early_stream=None, _single_rank_ok: bool = False) -> None:
    """C1 in three collectives so that the first overlaps the backward pass: ``flat[begin:end]`` (the gradient range
    ``beso_loss_grad_overlap`` completes first) is reduced from ``early_stream`` -- which the kernels' call has ordered
    behind the completion of that range -- while the current stream still runs the backward of the lower layers; the two
    remaining ranges follow on the current stream, which finally waits for the early collective.  RCCL executes each
    collective on its own stream behind the stream it was issued from, so issuing from ``early_stream`` is what lets it
    start early.  ``early_stream=None`` (CPU tensors / gloo): the same three reductions, in order."""
    if not is_distributed() and not (_single_rank_ok and dist.is_initialized()):      # (one-rank groups: the tests)
        return
    begin, end = early_range
    work = None
    if end > begin:
        if early_stream is not None:
            with torch.cuda.stream(early_stream):
                work = dist.all_reduce(flat[begin:end], op=dist.ReduceOp.SUM, async_op=True)
        else:
            dist.all_reduce(flat[begin:end], op=dist.ReduceOp.SUM)
    if begin > 0:
        dist.all_reduce(flat[:begin], op=dist.ReduceOp.SUM)
    if end < flat.numel():
        dist.all_reduce(flat[end:], op=dist.ReduceOp.SUM)
    if work is not None:
        work.wait()                                   # the current stream waits for the early collective


# ---------------------------------------------------------------------------------------------------------------------
# C1 as reduce-scatter + all-gather, with the optimizer step on the owned shard in between (ZeRO-1):
#   grads (flat, pre-scaled by 1/world)  --reduce-scatter-->  this rank's 1/world of the summed gradient
#   Adam(W) (+EMA) on that range of parameters only (1/world of the optimizer work and of the moment traffic)
#   updated parameter shards  --all-gather-->  every replica holds the new weights
# On the fully connected xGMI mesh both halves are direct exchanges that keep all seven links busy; the volume on the
# wire equals one all-reduce.  The EMA shadow is updated for the owned range only and gathered when it is next read.
# ---------------------------------------------------------------------------------------------------------------------
def shard_size(n: int, world: int) -> int:
    return (n + world - 1) // world


def shard_range_flat(n: int, world: int, rank_: int):
    """[lo, hi) of the flat parameter / gradient order owned by ``rank_`` (equal shards of ceil(n / world), the last
    one short)."""
    s = shard_size(n, world)
    return min(n, rank_ * s), min(n, (rank_ + 1) * s)


def reduce_scatter_flat(flat_padded: torch.Tensor, out_shard: torch.Tensor) -> None:
    """flat_padded: world * S floats (gradients, zero tail); out_shard (S floats) <- sum over ranks of this rank's shard."""
    dist.reduce_scatter_tensor(out_shard, flat_padded, op=dist.ReduceOp.SUM)


def all_gather_flat(full_padded: torch.Tensor, shard: torch.Tensor) -> None:
    """full_padded (world * S floats) <- the shards of all ranks, in rank order."""
    dist.all_gather_into_tensor(full_padded, shard)


class ShardedExchange:
    """Buffers and the two collectives of the sharded step for a parameter list of ``n`` floats in total."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params = list(params)
        self.n = sum(p.numel() for p in self.params)
        self.world, self.rank = world_size(), rank()
        self.S = shard_size(self.n, self.world)
        self.lo, self.hi = shard_range_flat(self.n, self.world, self.rank)
        dev = self.params[0].device
        self.padded = self.S * self.world
        self.g_shard = torch.zeros(self.S, dtype=torch.float32, device=dev)
        self.p_shard = torch.zeros(self.S, dtype=torch.float32, device=dev)
        self.full = torch.zeros(self.padded, dtype=torch.float32, device=dev)
        self._offs = []
        off = 0
        for p in self.params:
            self._offs.append(off)
            off += p.numel()

    def _pieces(self, lo, hi):
        """(parameter index, start in the tensor, stop in the tensor, start in [lo, hi)) of the tensors that intersect."""
        for i, (p, off) in enumerate(zip(self.params, self._offs)):
            a, b = max(lo, off), min(hi, off + p.numel())
            if a < b:
                yield i, a - off, b - off, a - lo

    def reduce_scatter_grads(self, flat_padded: torch.Tensor) -> None:
        """flat_padded[lo:hi] <- this rank's shard of the summed gradients (the rest of the buffer is left as it was)."""
        assert flat_padded.numel() == self.padded
        reduce_scatter_flat(flat_padded, self.g_shard)
        flat_padded[self.lo:self.hi].copy_(self.g_shard[: self.hi - self.lo])

    def all_gather_params(self) -> None:
        """Every replica's parameters <- the owners' updated shards."""
        with torch.no_grad():
            for i, a, b, at in self._pieces(self.lo, self.hi):
                self.p_shard[at:at + (b - a)].copy_(self.params[i].detach().reshape(-1)[a:b])
            all_gather_flat(self.full, self.p_shard)
            for p, off in zip(self.params, self._offs):
                p.copy_(self.full[off:off + p.numel()].view_as(p))

    def all_gather_flat_state(self, flat_state: torch.Tensor) -> None:
        """A flat per-parameter state (the EMA shadow) of which every rank holds its own range up to date: make it whole."""
        send = torch.zeros(self.S, dtype=torch.float32, device=flat_state.device)
        send[: self.hi - self.lo].copy_(flat_state[self.lo:self.hi])
        all_gather_flat(self.full, send)
        flat_state.copy_(self.full[: self.n])


def all_reduce_mean(x: torch.Tensor) -> torch.Tensor:
    """C3: mean over the ranks of a small tensor (the logged loss of equal-sized shards), in place."""
    if is_distributed():
        dist.all_reduce(x, op=dist.ReduceOp.SUM)
        x /= world_size()
    return x


def _parse_cpulist(text: str) -> List[int]:
    cpus: List[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def pin_to_gpu_numa_node(local_rank: int, sysfs: str = "/sys") -> Optional[int]:
    """One process per GPU: keep this rank's host threads (launch path, pinned staging buffers, the feed) on the NUMA node
    its GPU hangs off -- on a two-socket MI355X host half of the GPUs sit behind the other socket's xGMI / PCIe root.
    The node comes from the GPU's PCI function in sysfs, the CPUs from the node's cpulist; best effort: returns the node,
    or None when the topology cannot be read (single node, container without sysfs) and nothing is changed."""
    import os
    try:
        props = torch.cuda.get_device_properties(local_rank)
        bdf = f"{props.pci_domain_id:04x}:{props.pci_bus_id:02x}:{props.pci_device_id:02x}.0"
        node = int(open(os.path.join(sysfs, "bus", "pci", "devices", bdf, "numa_node")).read().strip())
        if node < 0:
            return None
        cpus = _parse_cpulist(open(os.path.join(sysfs, "devices", "system", "node", f"node{node}", "cpulist")).read())
        cpus = sorted(set(cpus) & set(os.sched_getaffinity(0)))
        if not cpus:
            return None
        os.sched_setaffinity(0, cpus)
        return node
    except Exception:      # noqa: BLE001   (a missing sysfs entry must not stop a run)
        return None


def job_mean(total: float, count: int, device=None) -> Optional[float]:
    """Mean of a host-side metric over the whole job: every rank contributes (sum, count) -- possibly (0, 0): an empty
    shard --; all ranks get the same value (None if nobody had anything).  A collective: every rank must call it.  What
    the training loops base early stopping and checkpoint decisions on, so that no rank leaves a loop, or enters the
    checkpoint's all-gather, alone."""
    if not is_distributed():
        return total / count if count else None
    dev = device if (device is not None and dist.get_backend() == "nccl") else "cpu"
    t = torch.tensor([float(total), float(count)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    s, n = t.tolist()
    return s / n if n > 0 else None


def broadcast_parameters(params: Iterable[torch.Tensor], src: int = 0) -> None:
    """C2: make every replica start from rank ``src``'s weights (one flat broadcast)."""
    if not is_distributed():
        return
    params = list(params)
    flat = torch.cat([p.detach().reshape(-1) for p in params])
    dist.broadcast(flat, src=src)
    off = 0
    with torch.no_grad():
        for p in params:
            n = p.numel()
            p.copy_(flat[off:off + n].view_as(p))
            off += n
