"""Data-parallel gradient reduction over RCCL / xGMI (new capability: the reference is single-GPU).

One process per GPU, per-GPU minibatch, identical initial weights (broadcast of the parameter arena), and a
bucketed sum all-reduce of the *gradient arena* (the flat float32 buffer FusedAdam owns) that overlaps with
backward: a bucket is launched on RCCL's stream the moment the last wgrad kernel that writes into it has been
enqueued.  Averaging (1/world) is folded into the Adam kernel's grad_scale, so no extra pass touches the
gradients.  Buckets are contiguous arena ranges, sized for xGMI (default 128 MiB: few, large collectives --
each GPU has 7 point-to-point links, a ring moves 2(N-1)/N of the payload over one link per hop).

Semantics: losses are batch means and every convolution / InstanceNorm layer is per-sample, so the average
of per-rank gradients equals the single-process gradient on the concatenated batch.  The ONE cross-sample coupling is
the BatchNorm2d inside the bottleneck-transformer blocks (``--n_blocks_attn_g > 0``, the default and configs[2]/[3]):
each rank normalises with its own batch statistics -- what torch's DistributedDataParallel does with a plain
``nn.BatchNorm2d`` -- so N ranks x batch 8 is not bit-equivalent to one rank x batch 8N there.  The running buffers
are broadcast from rank 0 at attach() time and then evolve per rank; ``sync_buffers(model)`` averages them across
ranks (call it before saving a checkpoint so rank 0's file does not carry rank-0-only statistics).
"""
from __future__ import annotations

import os
import sys

import torch
import torch.distributed as dist

from . import functional as Fh


class ArenaReducer:
    """Reduces `flat_g` (1-D tensor) across ranks in contiguous buckets.

    slices: [(param, offset, padded_numel)] in arena order.  writes_per_step: how many wgrad launches write each
    parameter's gradient during one backward (1 for G; 2 for D: the fake and the real pass)."""

    def __init__(self, flat_g, slices, writes_per_step=1, bucket_bytes=128 << 20, group=None, tail_bytes=None,
                 mode=None, g16_of=None):
        self.flat_g, self.group = flat_g, group
        # g16_of (round 6, "allreduce" mode under --fp16): callable param -> its float16 gradient arena view or None (FusedAdam
        # GRAD_F16: the weight-gradient kernel stores float16, the reference's own dtype for an autocast layer's gradient).  Those
        # ranges go over the links as they are -- 2 bytes per element with no cast pass (configs[3]: 92 % of the generator's 736 M
        # parameters are the trunk weights stored that way: 2.94 -> 1.59 GB per step and rank) -- and are SUMMED in float16: a sum that
        # leaves the float16 range becomes inf on every rank alike, the GradScaler's check on the reduced arena skips the step
        # everywhere and halves the scale, i.e. the loss scale settles up to log2(world) notches lower than on one GPU.
        self.g16_of = g16_of
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # "allreduce": one RCCL all-reduce per bucket (RCCL picks its own rings over the 7 xGMI links);
        # "rs_ag": reduce-scatter + all-gather per bucket -- the two halves of the same collective issued separately,
        # so each moves (N-1)/N of the bucket over all links at once on a fully connected node and the all-gather of
        # bucket i overlaps the reduce-scatter of bucket i+1 on RCCL's stream.  Needs a backend with
        # reduce_scatter_tensor (RCCL; not gloo), buckets are then padded to a multiple of the world size.
        # "sharded" (round 3, ZeRO-1 over the arena): reduce-scatter only; each rank owns 1/world of every bucket, runs the
        # inf / nan check and Adam on its shards alone (FusedAdam.shard), and the updated PARAMETER shards are all-gathered --
        # the same bytes on the links as one all-reduce, Adam's HBM stream divided by world (configs[3]: 22 GB -> 2.8 GB per
        # step and rank at world 8), m / v of the other ranks' shards never touched.
        self.mode = mode or os.environ.get("MDCTGAN_DDP_MODE", "allreduce")
        if self.mode not in ("allreduce", "rs_ag", "sharded"):
            raise ValueError("MDCTGAN_DDP_MODE must be allreduce, rs_ag or sharded")
        # MDCTGAN_DDP_GRAD_DTYPE=bf16 | f16 (opt-in, "allreduce" mode): the collective moves a 16-bit copy of the bucket -- half the
        # bytes on the xGMI links (configs[3]: 1.49 GB instead of 2.98 GB per step and rank).  The copy holds the MEAN (x 1/world
        # before the sum: a power of two, exact), so a float16 sum of `world` loss-scaled gradients overflows no earlier than a
        # single rank's gradient; the float32 arena receives mean x world, i.e. what the float32 sum would be up to the 16-bit
        # rounding of each rank's addend (bf16: 2^-9 relative, f16: 2^-11 -- the autocast gradients are float16-rounded products
        # already).  inf / nan survive both casts, and every rank sees the same reduced values, so the GradScaler's found_inf
        # check (FusedAdam.step, after finish()) takes the same decision everywhere.
        gd = os.environ.get("MDCTGAN_DDP_GRAD_DTYPE", "")
        if gd and gd not in ("bf16", "f16"):
            raise ValueError("MDCTGAN_DDP_GRAD_DTYPE must be bf16 or f16")
        if gd and self.mode != "allreduce":
            raise ValueError("MDCTGAN_DDP_GRAD_DTYPE needs MDCTGAN_DDP_MODE=allreduce")
        self.wire_dtype = {"bf16": torch.bfloat16, "f16": torch.float16}.get(gd)
        self.wire = torch.empty(flat_g.numel(), dtype=self.wire_dtype, device=flat_g.device) if self.wire_dtype else None
        self._unpack = []            # (lo, hi) of buckets whose reduced 16-bit copy finish() writes back
        self.writes_per_step = writes_per_step
        self.active = True
        self.force = dist.is_initialized()       # a 1-rank group still goes through RCCL (used to test the code path)
        self.buckets = []            # [lo, hi, n_params]
        self.param_bucket = {}
        self._members = []           # per bucket: [(param, offset, padded_numel)]
        self._piece_cache = {}
        # Backward fills the arena from its end, so the bucket that holds the FIRST parameters is the last one to
        # launch and the only one whose all-reduce cannot hide behind remaining backward work: keep it small
        # (tail_bytes, default a quarter of a bucket) and never let a large late-arena tensor ride in it.
        tail = max(1, (bucket_bytes // 4 if tail_bytes is None else tail_bytes) // 4)
        per_bucket = max(1, bucket_bytes // 4)
        lo, count, cur, writes = 0, 0, 0, 0
        self._bucket_writes = []           # wgrad launches a bucket waits for (a module applied k times writes k times)
        for p, off, n in slices:
            cap = tail if not self.buckets else per_bucket
            if count and off + n - lo > cap:                   # close before a tensor that would overflow the cap
                self.buckets.append([lo, off, count])
                self._bucket_writes.append(writes)
                lo, count, writes = off, 0, 0
            if len(self._members) == len(self.buckets):
                self._members.append([])
            self._members[-1].append((p, off, n))
            self.param_bucket[id(p)] = len(self.buckets)
            count += 1
            writes += int(getattr(p, "_mg_writes", 1)) * writes_per_step
            cur = off + n
        if count:
            self.buckets.append([lo, cur, count])
            self._bucket_writes.append(writes)
        self._reset()
        self._hook = Fh.register_grad_ready_hook(self._on_ready)

    def _reset(self):
        self.pending = list(self._bucket_writes)
        self.works = []

    def close(self):
        Fh.remove_grad_ready_hook(self._hook)

    def abort(self):
        """Forget a step that did not finish (a hipGraph capture that failed half way, Pix2PixHDModel.make_step): the handles of
        its collectives belong to the aborted capture, the bucket counters are mid-step."""
        self.works, self._unpack = [], []
        self._reset()

    def _split(self, lo, hi):
        """A bucket [lo, hi) is handed out in shards only over its prefix [lo, mid) whose length is a multiple of 8 * world:
        every shard then starts 16-byte aligned in the float32 arenas AND in the float16 shadow (the Adam / scaler kernels
        reject anything else: mg_adam_step_* `al16`), at every world size -- arena slices are padded to 8 elements only, so a
        plain n / world split is misaligned at world 4 and 8.  The tail [mid, hi) (< 8 * world elements) is
        all-reduced and updated redundantly by every rank."""
        q = 8 * self.world
        return lo + (hi - lo) // q * q

    def _pieces(self, i):
        """Bucket i as the tensors that travel: maximal runs of float32-stored gradients (views of flat_g) and of float16-stored ones
        (views of the optimiser's float16 gradient arena, same element offsets).  None: the whole bucket is float32."""
        if self.g16_of is None:
            return None
        halves = tuple(self.g16_of(p) is not None for p, _, _ in self._members[i])
        if not any(halves):
            return None
        hit = self._piece_cache.get(i)
        if hit is not None and hit[0] == halves:
            return hit[1]
        out, run = [], None          # run = [lo, hi, is_half, arena]
        for (p, off, n), h in zip(self._members[i], halves):
            arena = None
            if h:
                v = self.g16_of(p)
                arena = v._base if v._base is not None else v
            if run is not None and run[2] == h and run[1] == off and (not h or run[3] is arena):
                run[1] = off + n
            else:
                if run is not None:
                    out.append(run)
                run = [off, off + n, h, arena]
        out.append(run)
        pieces = [(a[3] if a[2] else self.flat_g)[a[0]:a[1]] for a in out]
        self._piece_cache[i] = (halves, pieces)
        return pieces

    def _launch(self, i):
        lo, hi, _ = self.buckets[i]
        if not (self.world > 1 or self.force) or not dist.is_initialized():     # (a reducer that outlived its group)
            return
        pieces = self._pieces(i)
        if pieces is not None:
            for t in pieces:
                self.works.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            return
        if self.wire is not None:
            w16 = self.wire[lo:hi]
            w16.copy_(self.flat_g[lo:hi])                 # round to nearest (inf / nan stay inf / nan)
            w16.mul_(1.0 / self.world)                    # the mean: exact for world = 2, 4, 8
            self.works.append(dist.all_reduce(w16, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            self._unpack.append((lo, hi))
            return
        mid = self._split(lo, hi) if self.mode in ("rs_ag", "sharded") else lo
        if mid > lo:
            shard = (mid - lo) // self.world
            rank = dist.get_rank(self.group)
            mine = self.flat_g[lo + rank * shard:lo + (rank + 1) * shard]
            w = dist.reduce_scatter_tensor(mine, self.flat_g[lo:mid], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            if self.mode == "sharded":
                self.works.append(w)
            else:
                self.works.append(dist.all_gather_into_tensor(self.flat_g[lo:mid], mine, group=self.group, async_op=True))
        if mid < hi:
            self.works.append(dist.all_reduce(self.flat_g[mid:hi], op=dist.ReduceOp.SUM, group=self.group,
                                              async_op=True))

    def _on_ready(self, p):
        if not self.active:
            return
        i = self.param_bucket.get(id(p))
        if i is None:
            return
        self.pending[i] -= 1
        if self.pending[i] == 0:
            self._launch(i)

    # -- sharded optimiser support (mode == "sharded") ---------------------------------------------------------
    def sharded(self):
        return self.mode == "sharded" and (self.world > 1 or self.force) and dist.is_initialized()

    def my_spans(self):
        """Arena ranges [lo, hi) whose reduced gradient this rank holds and whose parameters it updates: its 1/world of every
        bucket's shardable prefix, plus the bucket's all-reduced tail (every rank updates that redundantly)."""
        rank = dist.get_rank(self.group)
        out = []
        for lo, hi, _ in self.buckets:
            mid = self._split(lo, hi)
            shard = (mid - lo) // self.world
            if shard:
                out.append((lo + rank * shard, lo + (rank + 1) * shard))
            if mid < hi:
                out.append((mid, hi))
        return out

    def restrict(self, spans):
        """Intersect the optimiser's live spans with this rank's shards."""
        mine, out = self.my_spans(), []
        for a, b in spans:
            for lo, hi in mine:
                x, y = max(a, lo), min(b, hi)
                if x < y:
                    out.append((x, y))
        return out

    def agree(self, flag):
        """found_inf (0 / 1, one element on the device) of the GradScaler: a rank only sees its own shards, every rank must
        take the same skip / step decision (train.py:183-199 semantics on the global batch) -> MAX over the ranks."""
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)

    def gather(self, flat_p):
        """After the sharded update: every rank receives the other ranks' updated parameter shards.  Returns the pending
        collectives: the generator's all-gather is left running under the discriminator's backward pass and Adam step and
        waited for before the next forward (FusedAdam.finish_pending)."""
        rank = dist.get_rank(self.group)
        works = []
        for lo, hi, _ in self.buckets:
            mid = self._split(lo, hi)
            shard = (mid - lo) // self.world
            if shard:
                works.append(dist.all_gather_into_tensor(flat_p[lo:mid], flat_p[lo + rank * shard:lo + (rank + 1) * shard],
                                                         group=self.group, async_op=True))
        return works

    def finish(self):
        """Called before the optimiser step: flush buckets that never filled (frozen / unused parameters) and make
        the compute stream wait for every collective."""
        for i, left in enumerate(self.pending):
            if left > 0:
                self._launch(i)
        for w in self.works:
            w.wait()
        for lo, hi in self._unpack:                       # 16-bit wire format: back into the float32 arena as the SUM
            g = self.flat_g[lo:hi]
            g.copy_(self.wire[lo:hi])
            g.mul_(float(self.world))
        self._unpack = []
        self._reset()


def broadcast_arena(flat_p, src=0, group=None):
    """C2: identical initial weights on every rank."""
    if dist.is_initialized():
        dist.broadcast(flat_p, src=src, group=group)


SHARD_MIN_PARAMS = 64 << 20      # elements of one optimiser's arena from which MDCTGAN_DDP_MODE=auto picks "sharded" at world >= 4


def default_mode(world, n_params):
    """The reduction mode when MDCTGAN_DDP_MODE is unset: "allreduce", always.
    MDCTGAN_DDP_MODE=auto applies the size rule instead -- "sharded" (reduce-scatter, Adam on 1/world of the arena, all-gather of
    the updated parameters) when world >= 4 AND the arena holds >= 64 Mi parameters: both modes move the same bytes over xGMI, but
    the Adam kernel streams 28 B (float32) / 30 B (--fp16 shadow) per parameter and is a serial tail after backward -- configs[3]'s
    generator (736 M parameters) pays 4.6 ms of it per step and rank, 0.6 ms at world 8 when sharded.  The rule is NOT the default:
    the sharded path (found_inf MAX agreement, asynchronous parameter all-gather left running under the discriminator's backward
    pass, shadow re-sync) has only ever run over gloo and with a 1-rank RCCL group; until a world >= 4 RCCL run has shown loss and
    parameter equality against "allreduce" it stays opt-in.  Whatever the mode, attach() cross-checks a parameter
    checksum over the ranks after each of the first steps (MDCTGAN_DDP_CHECK_STEPS, default 3)."""
    env = os.environ.get("MDCTGAN_DDP_MODE")
    if env and env != "auto":
        return env
    if env == "auto":
        return "sharded" if (world >= 4 and n_params >= SHARD_MIN_PARAMS) else "allreduce"
    return "allreduce"


def check_replicas(flat_p, group=None, what="parameters"):
    """Cross-rank agreement of a parameter arena: every rank's (sum, sum of squares) in float64 must equal rank 0's bit for bit
    (identical weights, identical kernel, identical reduction order).  Raises on the first divergent step instead of letting the
    replicas drift apart silently."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    # float64 accumulation without a float64 copy of the arena (736 M parameters: 6 GB + 6 GB of temporaries otherwise)
    mine = torch.stack([flat_p.sum(dtype=torch.float64), torch.linalg.vector_norm(flat_p, ord=2, dtype=torch.float64)])
    ref = mine.clone()
    dist.broadcast(ref, src=0, group=group)
    # every rank raises together: [this rank differs from rank 0, this rank's checksum is not finite]
    flags = torch.stack([(mine != ref).any().float(), (~torch.isfinite(mine)).any().float()])
    dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=group)
    if float(flags[1]) != 0.0:      # nan != nan: do not report a blown-up run as divergence
        raise RuntimeError("data-parallel check: the %s checksum is not finite on at least one rank (rank %d holds %r): the "
                           "parameters contain inf / nan" % (what, dist.get_rank(group), mine.tolist()))
    if float(flags[0]) != 0.0:
        raise RuntimeError("data-parallel replicas diverged: the %s checksum of rank %d differs from rank 0's (%r vs %r)"
                           % (what, dist.get_rank(group), mine.tolist(), ref.tolist()))


def attach_optimizer(opt, writes_per_step=1, bucket_bytes=128 << 20, group=None):
    """Broadcast one FusedAdam's parameter arena from rank 0, create its gradient reducer and fold 1/world into the Adam
    kernel.  Returns the reducer (also used by Pix2PixHDModel.update_fixed_params when it replaces optimizer_G)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    slices = opt.arena_slices()
    broadcast_arena(opt.flat_p, 0, group)
    opt.resync_shadow()        # --fp16: the float16 shadow was cast from the PRE-broadcast weights
    n_params = sum(n for _, _, n in slices)
    mode = default_mode(world, n_params)
    # float16-stored gradients (FusedAdam GRAD_F16) travel as they are in "allreduce" mode; the shard arithmetic of the other
    # modes and the explicit 16-bit wire copy read the float32 arena alone.  MDCTGAN_DDP_NATIVE_G16=0: float32 arena, float32 wire.
    native16 = (mode == "allreduce" and not os.environ.get("MDCTGAN_DDP_GRAD_DTYPE", "")
                and os.environ.get("MDCTGAN_DDP_NATIVE_G16", "1") != "0")
    if hasattr(opt, "disable_g16") and not native16:
        opt.disable_g16()
    if world > 1 and (not dist.is_initialized() or dist.get_rank(group) == 0):
        print("[mdctgan_amd.ddp] %d ranks, %.1f M parameters: gradient reduction mode %r" % (world, n_params / 1e6, mode), file=sys.stderr, flush=True)
    red = ArenaReducer(opt.flat_g, slices, writes_per_step, bucket_bytes, group, mode=mode,
                       g16_of=(lambda p: getattr(p, "_mg_g16", None)) if native16 and hasattr(opt, "disable_g16") else None)
    red.bucket_bytes = bucket_bytes
    opt.grad_scale = 1.0 / world
    opt.pre_step_hook = red.finish
    opt.shard = red if red.sharded() else None
    return red


def _loose_tensors(model, arena_params):
    """Every parameter that is NOT in an optimiser arena (frozen / not optimised yet: --niter_fix_global) and every
    floating-point module buffer (BatchNorm running statistics)."""
    out = []
    for net in (getattr(model, "netG", None), getattr(model, "netD", None)):
        if net is None:
            continue
        out += [p.data for p in net.parameters() if id(p) not in arena_params]
        out += [b for b in net.buffers() if b.dtype.is_floating_point]
    return out


def sync_buffers(model, group=None):
    """Average the BatchNorm running statistics over ranks (see the module docstring)."""
    if not dist.is_initialized():
        return
    world = dist.get_world_size(group)
    for net in (getattr(model, "netG", None), getattr(model, "netD", None)):
        if net is None:
            continue
        for name, b in net.named_buffers():
            if b.dtype.is_floating_point:
                dist.all_reduce(b, op=dist.ReduceOp.SUM, group=group)
                b.div_(world)


def enable_sync_batchnorm(group=None):
    """SURVEY 8e opt-in: the BatchNorm2d layers of the bottleneck-attention blocks (configs[2] / [3]) normalise with the
    statistics of the WHOLE data-parallel batch -- N ranks x batch 8 then computes what one rank x batch 8N would -- at the
    price of two small all-reduces (slices x 2 x C doubles) per layer and pass.  Default off: per-rank statistics, what
    torch's DistributedDataParallel does with a plain nn.BatchNorm2d.  MDCTGAN_SYNC_BN=1 makes attach() call this."""
    from . import ops
    if not dist.is_initialized():
        raise RuntimeError("enable_sync_batchnorm needs an initialised process group")
    ops.SYNC_BN_GROUP = (group, dist.get_world_size(group))


def disable_sync_batchnorm():
    from . import ops
    ops.SYNC_BN_GROUP = None


def attach(model, bucket_bytes=None, group=None):
    """Wire a Pix2PixHDModel for data parallelism: broadcast both parameter arenas, every parameter outside them and
    every floating-point buffer from rank 0, create the G and D reducers and fold 1/world into the Adam kernels."""
    if bucket_bytes is None:       # MDCTGAN_DDP_BUCKET_MB: bucket size in MiB (default 128)
        bucket_bytes = int(os.environ.get("MDCTGAN_DDP_BUCKET_MB", "128")) << 20
    if os.environ.get("MDCTGAN_DDP_GRAD_DTYPE", "") == "f16" and getattr(model, "scaler", None) is None:
        # float16 on the wire overflows at 65504: safe only where a GradScaler checks the reduced arena before Adam (--fp16);
        # a float32 run would turn |g| > 65504 into inf silently (bf16 keeps float32's range)
        raise ValueError("MDCTGAN_DDP_GRAD_DTYPE=f16 needs --fp16 (a GradScaler that checks found_inf); use bf16 for float32 runs")
    reducers = {}
    d_writes = 1 if getattr(model, "stack_d_loss_passes", False) else 2    # wgrad launches per D parameter and step
    arena_params = set()
    for name, opt, writes in (("G", model.optimizer_G, 1), ("D", model.optimizer_D, d_writes)):
        reducers[name] = attach_optimizer(opt, writes, bucket_bytes, group)
        arena_params.update(id(p) for p, _, _ in opt.arena_slices())
    if dist.is_initialized():
        for t in _loose_tensors(model, arena_params):
            dist.broadcast(t, src=0, group=group)
    reducers["D"].active = False
    model.reducers = reducers
    model.ddp_check_steps = int(os.environ.get("MDCTGAN_DDP_CHECK_STEPS", "3"))      # see check_replicas
    if os.environ.get("MDCTGAN_SYNC_BN", "0") == "1" and dist.is_initialized():
        enable_sync_batchnorm(group)
    return reducers
