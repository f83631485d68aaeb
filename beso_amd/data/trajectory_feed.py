"""The training feed with the trajectories resident in HBM (second half).

The reference feeds ``train_step`` from ``DataLoader(TrajectorySlicerDataset(...), batch_size, shuffle=True)``
(kitchen_workspace_manager.py:149-155, trajectory_loader.py:77-197): one Python ``__getitem__`` per window, a collate,
a pinned copy and a ``.to(device)``.  That tops out at a few 10^4 windows per second per worker; the HIP training
step consumes 3 x 10^5 windows per second per GPU (1024 windows in 3.4 ms).  The datasets behind it are small padded
tensors (``TensorDataset(observations[N,T,obs], actions[N,T,act], masks[N,T])``: relay-kitchen 566 x 409 x 69
floats = 64 MB) -- a rounding error of 288 GB of HBM -- so this feed keeps them on the device and produces each dict
batch with ONE launch of ``beso_gather_windows``: a permutation drawn on the device, the slicer's (trajectory, start)
table, and the future-goal rule of ``__getitem__`` evaluated per sample inside the kernel.

``DeviceTrajectoryFeed`` iterates like the reference's train loader: ``len()`` batches per epoch, every window
exactly once per epoch when shuffling, the last short batch kept unless ``drop_last``; batches are dicts with the
keys ``train_step`` reads (``observation``, ``action``, ``goal_observation``), already on the device.  There is no
CPU implementation: on a machine without the HIP library construction fails.
"""
from __future__ import annotations

import ctypes as C
from typing import Iterator, Optional, Sequence

import numpy as np
import torch

from .. import _lib
from .. import distributed as bdist


def window_table(seq_lengths: Sequence[int], window: int):
    """The slicer's table (trajectory_loader.py:126-135): every (trajectory, start) with start + window <= length,
    trajectories in order, starts ascending; trajectories shorter than the window are skipped."""
    traj, start = [], []
    for i, n in enumerate(seq_lengths):
        n = int(n)
        if n - window >= 0:
            traj.append(np.full(n - window + 1, i, dtype=np.int32))
            start.append(np.arange(n - window + 1, dtype=np.int32))
    if not traj:
        return np.zeros(0, np.int32), np.zeros(0, np.int32)
    return np.concatenate(traj), np.concatenate(start)


class DeviceTrajectoryFeed:
    def __init__(self, observations, actions, seq_lengths, window: int, batch_size: int, device,
                 future_conditional: bool = False, min_future_sep: int = 0, future_seq_len: Optional[int] = None,
                 only_sample_tail: bool = False, only_sample_seq_end: bool = False, shuffle: bool = True,
                 drop_last: bool = False, seed: Optional[int] = None, rank: Optional[int] = None,
                 world_size: Optional[int] = None):
        if future_conditional and future_seq_len is None:
            raise AssertionError("must specify a future_seq_len")                      # trajectory_loader.py:115
        self.lib = _lib.load()                                                         # no library, no feed
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("DeviceTrajectoryFeed keeps the dataset in HBM: it needs a GPU device")
        f32 = lambda x: torch.as_tensor(x, dtype=torch.float32).to(self.device).contiguous()      # noqa: E731
        self.observations, self.actions = f32(observations), f32(actions)
        if self.observations.dim() != 3 or self.actions.dim() != 3 or self.observations.shape[:2] != self.actions.shape[:2]:
            raise ValueError("observations [N,T,obs] and actions [N,T,act] expected")
        self.n_traj, self.t_max, self.obs_dim = self.observations.shape
        self.act_dim = self.actions.shape[2]
        lengths = np.asarray([int(v) for v in seq_lengths], dtype=np.int32)
        if lengths.shape != (self.n_traj,) or lengths.min() < 0 or lengths.max() > self.t_max:
            raise ValueError("one valid length per trajectory, at most the padded length")
        self.window, self.batch_size = int(window), int(batch_size)
        self.goal_len = int(future_seq_len) if future_conditional else 0
        self.min_future_sep = int(min_future_sep)
        self.goal_mode = _lib.GOAL_TAIL if only_sample_tail else _lib.GOAL_SEQ_END if only_sample_seq_end else _lib.GOAL_RANDOM
        traj, start = window_table(lengths, self.window)
        if len(traj) == 0:
            raise ValueError(f"no trajectory is as long as the window ({self.window})")
        self.slices = np.stack([traj, start, start + self.window], axis=1)              # the reference's .slices, as rows
        self._seq_len = torch.from_numpy(lengths).to(self.device)
        self._traj = torch.from_numpy(traj).to(self.device)
        self._start = torch.from_numpy(start).to(self.device)
        self.shuffle, self.drop_last = bool(shuffle), bool(drop_last)
        # data-parallel ranks (default: the initialised process group) draw the SAME permutation (same seed -- rank 0's is
        # broadcast when none is given) and take interleaved shares of it (equal counts: the permutation is wrapped to a
        # multiple of the world size), so that one epoch of the job still sees every window; the goal draws differ per rank
        self.rank = bdist.rank() if rank is None else int(rank)
        self.world_size = bdist.world_size() if world_size is None else int(world_size)
        self._perm_gen = torch.Generator(device=self.device)
        self._draw_gen = torch.Generator(device=self.device)
        if seed is None:
            box = [int(torch.randint(0, 2 ** 31 - 1, (1,)).item())]
            if bdist.is_distributed() and world_size is None:
                torch.distributed.broadcast_object_list(box, src=0)
            seed = box[0]
        seed = int(seed)
        self._perm_gen.manual_seed(seed)
        self._draw_gen.manual_seed(seed * 1000003 + 17 + self.rank)

    @classmethod
    def from_sliced(cls, sliced, batch_size: int, device, **kw) -> "DeviceTrajectoryFeed":
        """From a ``TrajectorySlicerDataset`` (the reference's or any object with its attributes): the underlying
        trajectories are read once through ``sliced.dataset[i]`` / ``get_seq_length(i)`` and moved to the device."""
        base = sliced.dataset
        rows = [base[i] for i in range(len(base))]
        t_max = max(int(r[0].shape[0]) for r in rows)

        def stack(k):
            out = torch.zeros(len(rows), t_max, rows[0][k].shape[-1], dtype=torch.float32)
            for i, r in enumerate(rows):
                out[i, :r[k].shape[0]] = torch.as_tensor(r[k], dtype=torch.float32)
            return out
        feed = cls(stack(0), stack(1), [base.get_seq_length(i) for i in range(len(base))], sliced.window, batch_size, device,
                   future_conditional=sliced.future_conditional, min_future_sep=sliced.min_future_sep,
                   future_seq_len=sliced.future_seq_len, only_sample_tail=sliced.only_sample_tail,
                   only_sample_seq_end=getattr(sliced, "only_sample_seq_end", False), **kw)
        if [tuple(int(v) for v in s) for s in sliced.slices] != [tuple(int(v) for v in s) for s in feed.slices]:
            raise ValueError("the window table built here differs from sliced.slices")
        return feed

    # ------------------------------------------------------------------ sizes
    @property
    def n_windows(self) -> int:
        return len(self.slices)

    def _share(self) -> int:
        """Windows of one epoch per rank: the SAME count on every rank, ceil(n / world) -- the permutation is wrapped
        around to a multiple of the world size, as torch's DistributedSampler does.  Ranks that differed by one window
        could differ by one batch, and a rank with an extra batch would issue a gradient all-reduce nobody joins."""
        return (self.n_windows + self.world_size - 1) // self.world_size

    def __len__(self) -> int:
        n = self._share()
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    # ------------------------------------------------------------------ one batch
    def gather(self, window_ids: torch.Tensor, draws: Optional[torch.Tensor] = None) -> dict:
        """The dict batch of the windows ``window_ids`` (int64, device).  ``draws``: one non-negative int64 per sample
        for the random future-goal start (drawn here if omitted)."""
        ids = window_ids.to(device=self.device, dtype=torch.int64).contiguous()
        B = ids.numel()
        if self.goal_len > 0 and self.goal_mode == _lib.GOAL_RANDOM:
            if draws is None:
                draws = torch.randint(0, 2 ** 62, (B,), device=self.device, generator=self._draw_gen)
            draws = draws.to(device=self.device, dtype=torch.int64).contiguous()
            if draws.numel() != B:
                raise ValueError("one draw per sample expected")
        else:
            draws = None
        obs = torch.empty(B, self.window, self.obs_dim, device=self.device)
        act = torch.empty(B, self.window, self.act_dim, device=self.device)
        goal = torch.empty(B, self.goal_len, self.obs_dim, device=self.device) if self.goal_len > 0 else None
        with torch.cuda.device(self.device):
            st = self.lib.beso_gather_windows(
                self.observations.data_ptr(), self.actions.data_ptr(), self._seq_len.data_ptr(), self.n_traj, self.t_max,
                self.obs_dim, self.act_dim, self._traj.data_ptr(), self._start.data_ptr(), self.n_windows, ids.data_ptr(),
                draws.data_ptr() if draws is not None else None, B, self.window, self.goal_len, self.goal_mode,
                self.min_future_sep, obs.data_ptr(), act.data_ptr(), goal.data_ptr() if goal is not None else None,
                C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        _lib.check(st, "gather_windows")
        batch = {"observation": obs, "action": act}
        if goal is not None:
            batch["goal_observation"] = goal
        return batch

    # ------------------------------------------------------------------ one epoch
    def __iter__(self) -> Iterator[dict]:
        n = self.n_windows
        order = (torch.randperm(n, device=self.device, generator=self._perm_gen) if self.shuffle
                 else torch.arange(n, device=self.device))
        pad = self._share() * self.world_size - n
        if pad:
            order = torch.cat([order, order[:pad]])
        mine = order[self.rank::self.world_size]
        for k in range(len(self)):
            yield self.gather(mine[k * self.batch_size:(k + 1) * self.batch_size])
