"""Drop-in for ``beso.agents.diffusion_agents.beso_agent.BesoAgent`` (reference: beso_agent.py:28-598).

Same constructor kwargs (Hydra ``_target_`` swap), same public methods and externally mutated
attributes (``model``, ``sigma_min/max``, ``use_kde`` ...).  What changes underneath:

  * ``predict`` / ``evaluate`` evaluate the EMA weights through a packed kernel image of the EMA
    shadow that is refreshed only after ``ema_helper.update`` -- the reference clones all parameters,
    copies the shadow in and copies the originals back on EVERY call (beso_agent.py:343-345,380-381);
  * ``sample_loop`` hands ddim / euler / heun to the HIP library as one enqueue of all steps;
  * ``train_step`` all-reduces the score-matching gradients over the data-parallel ranks (RCCL over
    xGMI) before the optimizer step when a process group is initialised.
"""
import contextlib
import logging
import math
import os
from collections import deque
from functools import partial

import numpy as np
import torch
import torch.nn as nn

from ... import distributed as bdist
from ..._instantiate import instantiate
from ...networks.ema_helper.ema import ExponentialMovingAverage
from ...optim import FusedAdam, maybe_fuse
from ...data.prefetch import DevicePrefetcher
from ..base_agent import BaseAgent
from .k_diffusion import gc_sampling as ks
from .k_diffusion import utils
from .k_diffusion.classifier_free_sampler import ClassifierFreeSampleModel
from .k_diffusion.score_gpts import DiffusionGPT
from .k_diffusion.score_wrappers import GCDenoiser

log = logging.getLogger(__name__)

try:                      # logging only; absent on the GPU box
    import wandb
except ImportError:       # pragma: no cover
    wandb = None


def _wandb_log(payload):
    if wandb is not None and getattr(wandb, "run", None) is not None:
        wandb.log(payload)


class BesoAgent(BaseAgent):
    def __init__(self, model, input_encoder, optimization, device: str, obs_modalities: list,
                 goal_modalities: list, target_modality: str, max_train_steps: int, max_epochs: int,
                 train_method: str, eval_every_n_steps: int, use_ema: bool, goal_conditioned: bool,
                 pred_last_action_only: bool, rho: float, num_sampling_steps: int, lr_scheduler,
                 sampler_type: str, sigma_data: float, sigma_min: float, sigma_max: float,
                 sigma_sample_density_type: str, sigma_sample_density_mean: float,
                 sigma_sample_density_std: float, decay: float, update_ema_every_n_steps: int,
                 window_size: int, goal_window_size: int, use_kde: bool = False, patience: int = 10):
        super().__init__(model, input_encoder, optimization, obs_modalities, goal_modalities, target_modality,
                         device, max_train_steps, eval_every_n_steps, max_epochs)
        self.ema_helper = ExponentialMovingAverage(self.model.get_params(), decay, self.device)
        self.use_ema = use_ema
        # torch Adam / AdamW on a HIP device -> the one-launch fused step (same hyper-parameters, same
        # param_groups surface for the LR scheduler); any other optimizer class is left as configured
        self.optimizer = maybe_fuse(self.optimizer)
        self.lr_scheduler = instantiate(lr_scheduler, optimizer=self.optimizer)
        self.gc = goal_conditioned
        self.train_method = train_method
        self.epochs = max_epochs
        self.sampler_type = sampler_type
        self.num_sampling_steps = num_sampling_steps
        self.sigma_data = sigma_data
        self.sigma_min = sigma_min
        self.sigma_max = sigma_max
        self.rho = rho
        self.sigma_sample_density_type = sigma_sample_density_type
        self.sigma_sample_density_mean = sigma_sample_density_mean
        self.sigma_sample_density_std = sigma_sample_density_std
        self.decay = decay
        self.update_ema_every_n_steps = update_ema_every_n_steps
        self.patience = patience
        self.window_size = window_size
        self.goal_window_size = goal_window_size
        self.pred_last_action_only = pred_last_action_only
        # rolling contexts of the rollout (beso_agent.py:96-100)
        self.obs_context = deque(maxlen=self.window_size)
        self.goal_context = deque(maxlen=self.goal_window_size)
        self.action_context = deque(maxlen=self.window_size - 1)
        self.que_actions = True
        self.use_kde = use_kde
        self.noise_scheduler = 'exponential'
        # MI355X runtime state
        self._ema_packed = None
        self._ema_packed_key = None
        self._grad_bucket = None
        self._sharded_ex = None          # buffers of the sharded gradient exchange (BESO_AMD_C1=sharded)
        self._ema_partial = False        # sharded exchange: the EMA shadow is current for the owned range only
        self._train_graphs = {}          # batch shape -> captured forward+backward (train_step)
        self._train_graph_ok = True

    # ------------------------------------------------------------------ scaler / bounds
    def get_scaler(self, scaler):
        self.scaler = scaler

    def set_bounds(self, scaler):
        self.model.min_action = torch.from_numpy(scaler.y_bounds[0, :]).to(self.device)
        self.model.max_action = torch.from_numpy(scaler.y_bounds[1, :]).to(self.device)

    # ------------------------------------------------------------------ EMA evaluation scope
    def _hip_denoiser(self):
        """The beso_amd GCDenoiser behind ``self.model`` (which scripts may wrap in
        ClassifierFreeSampleModel: scripts/training.py:54-55), or None."""
        m = self.model
        if isinstance(m, ClassifierFreeSampleModel):
            m = m.model
        if isinstance(m, GCDenoiser) and isinstance(m.inner_model, DiffusionGPT):
            return m
        return None

    @contextlib.contextmanager
    def _ema_scope(self):
        """Evaluate with the EMA weights.  HIP model on the GPU: swap in the packed image of the
        shadow.  Anything else: the reference's store / copy_to / restore (beso_agent.py:343-381)."""
        # (sharded C1: a collective when the shadow is partial -- reached by every rank at the same step, with or without
        # use_ema, so that no rank enters it alone)
        self._complete_ema()
        if not self.use_ema:
            yield
            return
        den = self._hip_denoiser()
        first = next(iter(self.model.parameters()), None)        # (listing all 113 parameters costs 0.3 ms per call)
        if den is not None and first is not None and first.is_cuda:
            inner = den.inner_model
            key = (self.ema_helper.version, inner.precision, id(self.ema_helper))
            if self._ema_packed is None or self._ema_packed_key != key:
                self._ema_packed = inner.pack_external(self.ema_helper.shadow_params, into=self._ema_packed)
                self._ema_packed_key = key
            with inner.use_weights(self._ema_packed):
                yield
            return
        self.ema_helper.store(self.model.parameters())
        self.ema_helper.copy_to(self.model.parameters())
        if den is not None:
            den.inner_model.mark_weights_dirty()
        try:
            yield
        finally:
            self.ema_helper.restore(self.model.parameters())
            if den is not None:
                den.inner_model.mark_weights_dirty()

    # ------------------------------------------------------------------ training
    def _sync_replicas(self):
        """Data parallel: every replica starts from rank 0's weights (C2, one flat broadcast), and the EMA shadow is
        re-seeded from them (a shadow built from the pre-broadcast init would make the ranks' EMA weights differ)."""
        if bdist.is_distributed() and not getattr(self, "_replicas_synced", False):
            bdist.broadcast_parameters(self.model.get_params(), src=0)
            den = self._hip_denoiser()
            if den is not None:
                den.inner_model.mark_weights_dirty()
            self.ema_helper.load_shadow_params(self.model.get_params())
            self._ema_packed_key = None
            # every rank must draw its OWN noise, sigma, dropout and goal masks: scripts seed all ranks with
            # the same cfg.seed (scripts/training.py:27), so the per-process generators are moved apart by the rank
            seed = (torch.initial_seed() + 7919 * (bdist.rank() + 1)) % (2 ** 63 - 1)
            torch.manual_seed(seed)
            self._replicas_synced = True

    def _c1_mode(self) -> str:
        """How the data-parallel gradient exchange (C1) runs: 'overlap' (default: all-reduce in three ranges, the first
        under the backward of the lower layers), 'flat' (one all-reduce after the backward), 'sharded' (reduce-scatter,
        Adam(W) + EMA on the owned 1/world of the parameters, all-gather of the updated parameters: SURVEY.md 2.2)."""
        mode = os.environ.get("BESO_AMD_C1", "overlap")
        if os.environ.get("BESO_AMD_C1_OVERLAP", "1") == "0" and mode == "overlap":
            mode = "flat"
        if mode not in ("overlap", "flat", "sharded"):
            raise ValueError(f"BESO_AMD_C1={mode!r}: choose overlap, flat or sharded")
        return mode

    def _sharded(self):
        if getattr(self, "_sharded_ex", None) is None:
            self._sharded_ex = bdist.ShardedExchange(self.model.get_params())
        return self._sharded_ex

    def _complete_ema(self):
        """Sharded C1 updates the EMA shadow of the owned parameter range only; before the shadow is read (evaluation,
        checkpoint) the ranges are gathered -- a collective: every rank gets here at the same step."""
        if getattr(self, "_ema_partial", False) and bdist.is_distributed():
            self._sharded().all_gather_flat_state(self.ema_helper._flat)
            self.ema_helper.version += 1
        self._ema_partial = False

    def train_agent(self, train_loader, test_loader):
        self._sync_replicas()
        if self.train_method == 'epochs':
            self.train_agent_on_epochs(train_loader, test_loader, self.epochs)
        elif self.train_method == 'steps':
            self.train_agent_on_steps(train_loader, test_loader)
        else:
            raise ValueError('Either epochs or n_steps must be specified!')

    def train_agent_on_epochs(self, train_loader, test_loader, epochs):
        """Epoch mode with the reference loop's cadence (beso_agent.py:129-175): the logged / early-stopping test MSE is the
        LAST test batch's (the reference re-creates its list inside the loop, :138-142), the step counter advances once
        more per batch on top of train_step's own increment (:152) -- which shifts the EMA cadence
        `steps % update_ema_every_n_steps` -- and the LR scheduler gets an extra step whenever that counter hits a
        multiple of eval_every_n_steps (:153-154)."""
        best_test_mse, mean_mse, avg_test_mse = 1e10, 1e10, 1e10
        for epoch in range(epochs):
            # (the all-gather that completes a sharded EMA shadow is issued HERE, on every rank: a rank whose test loader
            # is empty never reaches evaluate() and would meet the others' all-gather with job_mean's all-reduce)
            self._complete_ema()
            test_mse = [self.evaluate(batch) for batch in test_loader]
            # data parallel: the ranks draw different evaluation noise (and may hold different test shards), so the
            # early-stopping / checkpoint decision is taken on the job-wide mean -- the same on every rank; a rank deciding
            # alone would leave the others inside the next gradient exchange
            last = bdist.job_mean(test_mse[-1] if test_mse else 0.0, 1 if test_mse else 0, self.device)
            if last is not None:
                mean_mse = avg_test_mse = last
            stop, best_test_mse = self.early_stopping(best_test_mse, mean_mse, self.patience, epochs)
            if stop:
                log.info('Early stopping!')
                break
            losses = []
            for batch in train_loader:
                losses.append(self.train_step(batch))
                self.steps += 1
                if self.steps % self.eval_every_n_steps == 0:
                    self.lr_scheduler.step()
                _wandb_log({"training/loss": losses[-1], "training/test_loss": avg_test_mse})
            _wandb_log({'training/epoch_loss': float(np.mean(losses)) if losses else 0.0,
                        'training/epoch_test_loss': avg_test_mse, 'training/epoch': epoch})
            log.info("Epoch %d: mean test mse %s, mean train loss %s", epoch, avg_test_mse,
                     float(np.mean(losses)) if losses else None)
        self.store_model_weights(self.working_dir)
        log.info("Training done!")

    def train_agent_on_steps(self, train_loader, test_loader):
        """max_train_steps optimizer steps; every eval_every_n_steps: test MSE of the sampler and a
        checkpoint on improvement (beso_agent.py:177-213)."""
        best_test_mse, avg_test_mse = 1e10, 1e10
        # next batch is copied host -> device on a side stream while the current step runs
        train_loader = DevicePrefetcher(train_loader, self.device)
        stream = iter(train_loader)
        for step in range(self.max_train_steps):
            if not self.steps % self.eval_every_n_steps:
                self._complete_ema()          # on every rank, test batches or not (see train_agent_on_epochs)
                scores = [self.evaluate(batch) for batch in test_loader]
                # (job-wide mean: every rank takes the same checkpoint decision -- store_model_weights holds a collective)
                mean = bdist.job_mean(sum(scores), len(scores), self.device)
                if mean is not None:
                    avg_test_mse = mean
                log.info("Step %d: Mean test mse is %s", step, avg_test_mse)
                if avg_test_mse < best_test_mse:
                    best_test_mse = avg_test_mse
                    self.store_model_weights(self.working_dir)
                    log.info('New best test loss. Stored weights have been updated!')
            try:
                batch = next(stream)
            except StopIteration:
                stream = iter(train_loader)
                batch = next(stream)
            loss = self.train_step(batch)
            if not self.steps % 1000:
                log.info("Step %d: Mean batch loss mse is %s", step, loss)
            _wandb_log({"loss": loss, "test_loss": avg_test_mse})
        self.store_model_weights(self.working_dir)
        log.info("Training done!")

    # ------------------------------------------------------------------ training step internals
    def _loss_backward(self, state, action, goal):
        """noise ~ N(0, I), sigma ~ the configured density, score-matching loss, backward (beso_agent.py:226-235)."""
        noise = torch.randn_like(action)
        sigma = self.make_sample_density()(shape=(len(action),), device=self.device)
        step = self.model.hip_train_step(state, action, goal, noise, sigma) if hasattr(self.model, "hip_train_step") else None
        if step is not None:
            # HIP forward + backward (beso_loss_grad): gradients land in views of one flat buffer, already
            # divided by the world size for the data-parallel mean
            self.optimizer.zero_grad(set_to_none=True)
            self._hip_step = step
            # under data parallelism the all-reduce of the upper layers' gradients starts under the backward of the lower
            # ones: the call orders self._c1_stream behind the completion of that range (BESO_AMD_C1_OVERLAP=0: one flat
            # all-reduce after the backward)
            early = None
            if bdist.is_distributed() and self._c1_mode() == "sharded":
                step.pad_to = self._sharded().padded
            if bdist.is_distributed() and state.is_cuda and self._c1_mode() == "overlap" \
                    and not torch.cuda.is_current_stream_capturing():
                if getattr(self, "_c1_stream", None) is None:
                    self._c1_stream = torch.cuda.Stream(state.device)
                early = self._c1_stream
            self._c1_early = early
            # the loss value is final a third of the way into the step: a side stream is released at that point, and
            # train_step reads the loss there -- its `loss.item()` (beso_agent.py:248) then returns while the backward pass
            # and the optimizer are still running, and the host prepares the next step under them (BESO_AMD_ASYNC_LOSS=0:
            # the read waits for the whole step, as a plain .item() on the compute stream does)
            self._loss_ready = None
            if state.is_cuda and os.environ.get("BESO_AMD_ASYNC_LOSS", "1") != "0" and not torch.cuda.is_current_stream_capturing():
                if getattr(self, "_loss_stream", None) is None:
                    self._loss_stream = torch.cuda.Stream(state.device)
                self._loss_ready = self._loss_stream
            # (goal masking for classifier-free guidance, mask_cond, is applied by the kernel from cond_mask_prob)
            return step.loss_backward(state, action, goal, noise, sigma, grad_scale=1.0 / bdist.world_size(),
                                      early_stream=early, loss_stream=self._loss_ready)
        self._hip_step = None
        loss = self.model.loss(state, action, goal, noise, sigma)
        self.optimizer.zero_grad()
        loss.backward()
        return loss

    def _use_train_graph(self, state) -> bool:
        # opt-in: measured 16.6 vs 17.4 ms per 1024-sample step -- the eager step is bound by its fp32 kernels,
        # not by launches
        if not (state.is_cuda and self._train_graph_ok and os.environ.get("BESO_AMD_TRAIN_GRAPH", "0") == "1"):
            return False
        # the dropout / goal-mask seed of the HIP step is a host scalar: a captured graph would replay one mask forever, so
        # models with any dropout or cond_mask_prob > 0 are not captured at all (no warm-up steps spent on a doomed capture)
        den = self._hip_denoiser()
        if den is not None:
            inner = den.inner_model
            if any(float(p) > 0.0 for p in getattr(inner, "_pdrops", ())) or float(getattr(inner, "cond_mask_prob", 0.0) or 0.0) > 0.0:
                self._train_graph_ok = False
                log.info("BESO_AMD_TRAIN_GRAPH=1 ignored: the model has dropout / cond_mask_prob > 0 (host-drawn seed)")
                return False
        return True

    def _graphed_loss_backward(self, state, action, goal):
        """The same forward + backward replayed as ONE HIP graph per batch shape.  The eager training step
        issues several thousand small kernels; the graph removes the launch gaps and changes no arithmetic
        (opt-in, BESO_AMD_TRAIN_GRAPH=1: the measured gain is 5 %).  Inputs are copied into static
        buffers, noise / sigma / dropout masks are drawn inside the graph from torch's graph-safe generator,
        gradients land in static .grad tensors that the (fused) optimizer reads afterwards."""
        key = (tuple(state.shape), tuple(action.shape), None if goal is None else tuple(goal.shape))
        g = self._train_graphs.get(key)
        if g is None:
            g = dict(state=state.clone(), action=action.clone(), goal=None if goal is None else goal.clone())

            def fwd_bwd():
                noise = torch.randn_like(g["action"])
                sigma = self.make_sample_density()(shape=(len(g["action"]),), device=self.device)
                loss = self.model.loss(g["state"], g["action"], g["goal"], noise, sigma)
                loss.backward()
                return loss

            cur = torch.cuda.current_stream()
            side = torch.cuda.Stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):                    # warm-up off the capture stream (allocator, autotuning)
                for _ in range(3):
                    self.optimizer.zero_grad(set_to_none=True)
                    fwd_bwd()
            cur.wait_stream(side)
            self.optimizer.zero_grad(set_to_none=True)       # the captured backward then WRITES (not accumulates) .grad
            graph = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(graph):
                    g["loss"] = fwd_bwd()
            except RuntimeError as e:                        # e.g. a sigma density that copies from the host
                log.warning("training step could not be captured into a HIP graph (%s); running it eagerly", e)
                self._train_graph_ok = False
                self._train_graphs = {}
                return self._loss_backward(state, action, goal)
            g["graph"] = graph
            self._train_graphs = {key: g}                    # one shape at a time: grads belong to the latest graph
            graph.replay()                                   # capture records, it does not execute
        else:
            g["state"].copy_(state)
            g["action"].copy_(action)
            if goal is not None:
                g["goal"].copy_(goal)
            g["graph"].replay()
        return g["loss"]

    def train_step(self, batch: dict):
        """One score-matching step (beso_agent.py:215-248): noise ~ N(0, I), sigma ~ the configured
        density, loss = GCDenoiser.loss, optimizer + LR scheduler + EMA.  Under data parallelism the
        gradients are averaged over the ranks first.

        Completion semantics (BESO_AMD_ASYNC_LOSS=1, the default): the returned float is this step's loss, but the call
        returns when the FORWARD half is done -- the backward pass and the optimizer of this step are still queued on the
        compute stream.  Consequences: (i) a device fault in this step's backward / optimizer surfaces at the next
        synchronising call (the next step's loss read, `evaluate`, a checkpoint), not here; (ii) timing `train_step()` per
        call without `torch.cuda.synchronize()` measures the forward half, not the step; (iii) `self._loss_stream` belongs
        to the step (the library also runs the next step's weight copies on it): nothing else may be queued there.
        Everything later on the compute stream is ordered behind the update as usual; BESO_AMD_ASYNC_LOSS=0 restores the
        reference's full synchronisation per step."""
        state, action, goal = self.process_batch(batch, predict=False)
        if not self.model.training:          # (.train() / .eval() set the whole tree: the root's flag tells; walking the 69 modules
            self.model.train()               #  costs 0.1 ms per step)
        self.model.training = True
        self._c1_early = None
        self._loss_ready = None
        if self._use_train_graph(state):
            loss = self._graphed_loss_backward(state, action, goal)
        else:
            loss = self._loss_backward(state, action, goal)
        shard = None
        loss_stream = getattr(self, "_loss_ready", None)
        if bdist.is_distributed() and loss_stream is not None:
            # C3 first: the global-batch mean of the loss is exchanged on the loss stream, in front of the gradients' exchange
            # (every rank issues its collectives in this order), so that it waits for the forward half only
            with torch.cuda.stream(loss_stream):
                loss = bdist.all_reduce_mean(loss.detach().clone())
        if bdist.is_distributed():
            flat = self._hip_step.flat_grads() if getattr(self, "_hip_step", None) is not None else None
            sharded = self._c1_mode() == "sharded"
            if flat is not None and sharded and isinstance(self.optimizer, FusedAdam):
                # reduce-scatter: this rank ends up with the summed gradient of the parameter range it owns
                ex = self._sharded()
                ex.reduce_scatter_grads(self._hip_step.flat_grads_padded())
                shard = (ex.lo, ex.hi)
            elif flat is not None and getattr(self, "_c1_early", None) is not None:
                bdist.all_reduce_sum_overlapped(flat, self._hip_step.early_range(), self._c1_early)
            elif flat is not None:
                bdist.all_reduce_sum(flat)                    # C1 on the flat buffer the kernels wrote (pre-scaled)
            else:
                if self._grad_bucket is None:
                    self._grad_bucket = bdist.GradientBucket(self.model.get_params())
                self._grad_bucket.sync(decomposed=sharded)   # (eager optimizer: the same exchange as its two halves)
        self.steps += 1
        do_ema = self.steps % self.update_ema_every_n_steps == 0
        if isinstance(self.optimizer, FusedAdam):
            # Adam(W) over all tensors and the EMA of the updated parameters in ONE HIP launch -- over the owned range
            # only in the sharded exchange, followed by the all-gather of the updated parameters
            self.optimizer.step(ema=self.ema_helper if do_ema else None, shard=shard)
            if shard is not None:
                self._sharded().all_gather_params()
                self._ema_partial = getattr(self, "_ema_partial", False) or do_ema
            self.lr_scheduler.step()
        else:
            self.optimizer.step()
            self.lr_scheduler.step()
            if do_ema:
                self.ema_helper.update(self.model.parameters())
        if loss_stream is not None:
            with torch.cuda.stream(loss_stream):
                return loss.item()
        if bdist.is_distributed():
            loss = bdist.all_reduce_mean(loss.detach().clone())      # C3: the logged loss is the global-batch mean
        return loss.item()

    @torch.no_grad()
    def evaluate(self, batch: dict):
        """MSE between sampled and ground-truth action windows; always the exponential schedule
        (beso_agent.py:250-289)."""
        state, action, goal = self.process_batch(batch, predict=True)
        with self._ema_scope():
            self.model.eval()
            self.model.training = False
            sigmas = ks.get_sigmas_exponential(self.num_sampling_steps, self.sigma_min, self.sigma_max, 'cpu')
            x = torch.randn_like(action) * self.sigma_max
            x_0 = self.sample_loop(sigmas, x, state, goal, self.sampler_type)
            if self.pred_last_action_only:
                x_0 = x_0.reshape(x_0.shape[0], 1, -1)
            mse = nn.functional.mse_loss(x_0, action, reduction="none").mean().item()
        return mse

    # ------------------------------------------------------------------ rollout inference
    def reset(self):
        self.obs_context.clear()
        self.action_context.clear()

    @torch.no_grad()
    def predict(self, batch: dict, new_sampler_type=None, get_mean=None, new_sampling_steps=None,
                extra_args=None, noise_scheduler=None) -> torch.Tensor:
        """One environment step (beso_agent.py:296-388): append the observation to the window,
        draw x_T for the newest action, denoise the WHOLE action window (previous actions + fresh
        noise), keep the last action, clip, un-scale, remember it."""
        noise_scheduler = self.noise_scheduler if noise_scheduler is None else noise_scheduler
        state, goal, _ = self.process_batch(batch, predict=True)
        if state.dim() == 2 and self.window_size > 1:
            self.obs_context.append(state)
            input_state = torch.stack(tuple(self.obs_context), dim=1)
        else:
            input_state = state
        if goal.dim() == 2 and self.window_size > 1:
            goal = goal.unsqueeze(0)                                       # 'b d -> 1 b d'
        sampler_type = self.sampler_type if new_sampler_type is None else new_sampler_type
        n_steps = self.num_sampling_steps if new_sampling_steps is None else new_sampling_steps
        act_dim = self.scaler.y_bounds.shape[1]
        with self._ema_scope():
            if self.model.training:                          # (module.eval() walks ~100 submodules: 0.35 ms per call)
                self.model.eval()
            sigmas = self.get_noise_schedule(n_steps, noise_scheduler)
            n = len(input_state) * (get_mean if get_mean is not None else 1)
            if self.window_size > 1 or get_mean is None:
                x = torch.randn((n, 1, act_dim), device=self.device) * self.sigma_max
                if self.window_size > 1 and get_mean is None and len(self.action_context) > 0:
                    x = torch.cat((*self.action_context, x), dim=1)        # (= cat([cat(context), x]): one launch)
            else:
                x = torch.randn((n, act_dim), device=self.device) * self.sigma_max
            x_0 = self.sample_loop(sigmas, x, input_state, goal, sampler_type, extra_args)
            if x_0.dim() == 3 and x_0.size(1) > 1:
                x_0 = x_0[:, -1, :]
            x_0 = self.scaler.clip_action(x_0)
        model_pred = self.scaler.inverse_scale_output(x_0)
        if model_pred.dim() == 2:
            x_0 = x_0.unsqueeze(1)
        self.action_context.append(x_0)
        return model_pred

    def sample_loop(self, sigmas, x_t: torch.Tensor, state: torch.Tensor, goal: torch.Tensor, sampler_type: str,
                    extra_args={}):
        """Dispatch on ``sampler_type`` (beso_agent.py:390-456).  Only 'heun' receives s_churn / s_min;
        a non-empty ``extra_args`` must carry both 's_churn' and 'keep_last_actions' (KeyError
        otherwise, as in the reference :408-410)."""
        extra_args = {} if extra_args is None else extra_args
        s_churn = extra_args.get('s_churn', 0)
        s_min = extra_args.get('s_min', 0)
        reduced = {k: extra_args[k] for k in ('s_churn', 'keep_last_actions')} if extra_args else {}
        scaler = self.scaler if extra_args.get('use_scaler', False) else None
        m = self.model
        if sampler_type == 'lms':
            return ks.sample_lms(m, state, x_t, goal, sigmas, scaler=scaler, disable=True, extra_args=reduced)
        if sampler_type == 'heun':
            return ks.sample_heun(m, state, x_t, goal, sigmas, scaler=scaler, s_churn=s_churn, s_tmin=s_min,
                                  disable=True)
        if sampler_type == 'euler':
            return ks.sample_euler(m, state, x_t, goal, sigmas, scaler=scaler, disable=True)
        if sampler_type == 'ancestral':
            return ks.sample_dpm_2_ancestral(m, state, x_t, goal, sigmas, scaler=scaler, disable=True)
        if sampler_type == 'euler_ancestral':
            return ks.sample_euler_ancestral(m, state, x_t, goal, sigmas, scaler=scaler, disable=True)
        if sampler_type == 'dpm':
            return ks.sample_dpm_2(m, state, x_t, goal, sigmas, disable=True)
        if sampler_type == 'ddim':
            return ks.sample_ddim(m, state, x_t, goal, sigmas, scaler=scaler, disable=True)
        if sampler_type == 'dpm_adaptive':
            return ks.sample_dpm_adaptive(m, state, x_t, goal, sigmas[-2].item(), sigmas[0].item(), disable=True)
        if sampler_type == 'dpm_fast':
            return ks.sample_dpm_fast(m, state, x_t, goal, sigmas[-2].item(), sigmas[0].item(), len(sigmas),
                                      disable=True)
        if sampler_type == 'dpmpp_2s_ancestral':
            return ks.sample_dpmpp_2s_ancestral(m, state, x_t, goal, sigmas, scaler=scaler, disable=True)
        if sampler_type == 'dpmpp_2s':
            return ks.sample_dpmpp_2s(m, state, x_t, goal, sigmas, scaler=scaler, disable=True)
        if sampler_type == 'dpmpp_2m':
            return ks.sample_dpmpp_2m(m, state, x_t, goal, sigmas, scaler=scaler, disable=True)
        if sampler_type == 'dpmpp_2m_sde':
            return ks.sample_dpmpp_sde(m, state, x_t, goal, sigmas, scaler=scaler, disable=True)
        raise ValueError('desired sampler type not found!')

    # ------------------------------------------------------------------ checkpoints
    def load_pretrained_model(self, weights_path: str, **kwargs) -> None:
        """``model_state_dict.pth`` holds the EMA weights (beso_agent.py:458-464); the EMA shadow is
        re-seeded from the loaded parameters."""
        state = torch.load(os.path.join(weights_path, "model_state_dict.pth"), map_location=self.device)
        self.model.load_state_dict(state)
        self.ema_helper = ExponentialMovingAverage(self.model.get_params(), self.decay, self.device)
        self._ema_packed_key = None
        log.info('Loaded pre-trained model parameters')

    def store_model_weights(self, store_path: str) -> None:
        """EMA weights -> model_state_dict.pth, raw weights -> non_ema_model_state_dict.pth
        (beso_agent.py:466-476)."""
        self._complete_ema()
        if bdist.rank() != 0:
            return
        raw = {k: v.detach().clone() for k, v in self.model.state_dict().items()}
        ema = dict(raw)
        if self.use_ema:
            names = [n for n, p in self.model.named_parameters() if p.requires_grad]
            for n, s in zip(names, self.ema_helper.shadow_params):
                ema[n] = s.detach().clone()
        torch.save(ema, os.path.join(store_path, "model_state_dict.pth"))
        torch.save(raw, os.path.join(store_path, "non_ema_model_state_dict.pth"))

    # ------------------------------------------------------------------ sigma density / schedules
    def make_sample_density(self):
        """Training distribution of sigma (beso_agent.py:541-580)."""
        kind = self.sigma_sample_density_type
        if kind == 'lognormal':
            return partial(utils.rand_log_normal, loc=self.sigma_sample_density_mean,
                           scale=self.sigma_sample_density_std)
        if kind == 'loglogistic':
            return partial(utils.rand_log_logistic, loc=math.log(self.sigma_data), scale=0.5,
                           min_value=self.sigma_min, max_value=self.sigma_max)
        if kind == 'loguniform':
            return partial(utils.rand_log_uniform, min_value=self.sigma_min, max_value=self.sigma_max)
        if kind == 'uniform':
            return partial(utils.rand_uniform, min_value=self.sigma_min, max_value=self.sigma_max)
        if kind == 'v-diffusion':
            return partial(utils.rand_v_diffusion, sigma_data=self.sigma_data, min_value=self.sigma_min,
                           max_value=self.sigma_max)
        if kind == 'discrete':
            return partial(utils.rand_discrete, values=self.get_noise_schedule(self.num_sampling_steps, 'exponential'))
        raise ValueError('Unknown sample density type')

    def get_noise_schedule(self, n_sampling_steps, noise_schedule_type):
        """(beso_agent.py:582-598).  Schedules are built on the host: the samplers read them there."""
        if noise_schedule_type == 'karras':
            return ks.get_sigmas_karras(n_sampling_steps, self.sigma_min, self.sigma_max, self.rho, 'cpu')
        if noise_schedule_type == 'exponential':
            return ks.get_sigmas_exponential(n_sampling_steps, self.sigma_min, self.sigma_max, 'cpu')
        if noise_schedule_type == 'vp':
            return ks.get_sigmas_vp(n_sampling_steps, device='cpu')
        if noise_schedule_type == 'linear':
            return ks.get_sigmas_linear(n_sampling_steps, self.sigma_min, self.sigma_max, device='cpu')
        if noise_schedule_type == 'cosine_beta':
            return ks.cosine_beta_schedule(n_sampling_steps, device='cpu')
        if noise_schedule_type == 've':
            return ks.get_sigmas_ve(n_sampling_steps, self.sigma_min, self.sigma_max, device='cpu')
        if noise_schedule_type == 'iddpm':
            return ks.get_iddpm_sigmas(n_sampling_steps, self.sigma_min, self.sigma_max, device='cpu')
        raise ValueError('Unknown noise schedule type')
