"""Data parallelism for the VAE train step: one process per GPU over torch.distributed.

The reference is single-device.  Samples are independent except through train-mode
BatchNorm, and the ELBO is a SUM over samples (train.py:41-48), so a step on W equal batch shards
equals the single-GPU step on the global batch when

  * every training-mode BatchNorm uses the statistics of the global batch, forward and backward
    (the executor all-gathers one row of per-channel sums per rank through `vae_set_collective`), and
  * the flat gradient buffer is all-reduced with SUM (not mean) before Adam.

Parameters, Adam state and BatchNorm running statistics are replicated and stay bit-identical across
ranks because every rank applies the same update.  eps is drawn per rank from a rank-offset seed.
backend "nccl" is RCCL on ROCm; "gloo" (CPU tensors, or GPU tensors staged through the host) serves
the CPU tests and single-GPU rehearsals.
"""
import ctypes

import torch
import torch.distributed as dist

from . import _lib as L
from .train import FusedTrainStep


def shard_bounds(global_batch, rank, world):
    """[lo, hi) of the contiguous shard of `rank`; the first global_batch % world ranks get one more."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"bad rank {rank} / world {world}")
    q, r = divmod(global_batch, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


class Collective:
    """The two exchanges of the data-parallel step on a torch.distributed group."""

    def __init__(self, group=None):
        if not dist.is_initialized():
            raise RuntimeError("vae_amd.parallel needs torch.distributed.init_process_group() first")
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)

    def _staged(self, t):
        return t.is_cuda and self.backend == "gloo"

    def all_gather_rows(self, rows):
        """rows: contiguous (world, n) tensor whose row `rank` is filled; fills the others in place."""
        if rows.shape[0] != self.world or not rows.is_contiguous():
            raise ValueError(f"all_gather_rows: expected a contiguous ({self.world}, n) tensor, got {tuple(rows.shape)}")
        if self._staged(rows):
            mine = rows[self.rank].cpu()
            out = torch.empty(rows.numel(), dtype=rows.dtype)
            dist.all_gather_into_tensor(out, mine.view(-1), group=self.group)
            rows.copy_(out.view(rows.shape))
        elif rows.is_cuda:
            dist.all_gather_into_tensor(rows.view(-1), rows[self.rank].view(-1), group=self.group)  # in place (RCCL allows it)
        else:
            mine = rows[self.rank].clone()
            dist.all_gather_into_tensor(rows.view(-1), mine.view(-1), group=self.group)
        return rows

    def all_reduce_sum_async(self, t):
        """Starts the in-place SUM of `t` and returns a handle whose wait() orders the current stream after it.
        On RCCL the collective runs on the process group's own stream, behind the work already enqueued on the
        current stream, so kernels enqueued next overlap with it.  Staged/CPU groups reduce synchronously."""
        if t.is_cuda and self.backend != "gloo":
            return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.all_reduce_sum(t)
        return None

    def all_reduce_sum(self, t):
        """In-place SUM over the ranks (the loss is reduction='sum', so gradients add, not average)."""
        if self._staged(t):
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t


def rccl_path():
    """The librccl that torch.distributed's nccl backend uses in this process (torch ships its own copy)."""
    import os
    p = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    return p if os.path.exists(p) else ""


def _all_ranks_ok(ok, group, device=None):
    """MIN over the ranks of a success flag: every rank learns whether ALL of them succeeded."""
    t = torch.tensor([1 if ok else 0], dtype=torch.int32)
    if dist.get_backend(group) != "gloo":
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(int(t.cpu()[0]))


def native_init(group=None, device=None, required=True, single_comm=None):
    """Joins the library's own RCCL communicators (include/vae_amd.h: vae_dp_*) with the ranks of `group`; rank 0's
    id travels over torch.distributed.  Idempotent per process.  Returns the world size.

    The outcome is agreed on COLLECTIVELY, so that a failure on one rank cannot desynchronise the others: every rank
    makes the same sequence of torch.distributed calls whatever happens locally -- (1) every rank opens librccl and
    draws an id (only rank 0's is used; the call doubles as the "can this rank load RCCL" probe), (2) the id is
    broadcast, (3) the success flags are reduced with MIN, (4) only if every rank succeeded do the ranks enter the
    collective vae_dp_init, (5) whose result is reduced the same way.  When some rank failed: required=True raises
    VaeAmdError on EVERY rank, required=False returns 0 on every rank (they fall back together).
    single_comm: True = one communicator and one stream for every collective (vae_dp_set_single_comm); None reads the
    environment variable VAE_DP_SINGLE_COMM (default 0: two communicators, gradient buckets overlapped)."""
    import os
    lib = L.lib()
    if lib.vae_dp_world() > 0:
        return lib.vae_dp_world()
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    gloo = dist.get_backend(group) == "gloo"
    dev = None
    if not gloo or device is not None:
        dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        if dev.type == "cuda" and dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
    if single_comm is None:
        single_comm = os.environ.get("VAE_DP_SINGLE_COMM", "0") not in ("", "0")
    path = rccl_path().encode()
    ident = (ctypes.c_ubyte * L.DP_ID_BYTES)()
    err = None
    rc = lib.vae_dp_set_single_comm(1 if single_comm else 0)
    if rc == 0:
        rc = lib.vae_dp_unique_id(path, ident)
    if rc != 0:
        err = f"vae_dp_unique_id failed ({rc}): {lib.vae_last_error().decode()}"
    t = torch.tensor(list(ident), dtype=torch.uint8)
    t = t if gloo else t.to(dev)
    dist.broadcast(t, src=0, group=group)
    ok = _all_ranks_ok(err is None, group, dev)
    if ok:
        ident = (ctypes.c_ubyte * L.DP_ID_BYTES)(*t.cpu().tolist())
        with torch.cuda.device(dev):
            rc = lib.vae_dp_init(path, ident, rank, world)
        if rc != 0:
            err = f"vae_dp_init failed ({rc}): {lib.vae_last_error().decode()}"
        ok = _all_ranks_ok(err is None, group, dev)
        if not ok:
            lib.vae_dp_shutdown()  # the ranks that did join leave again: nobody keeps half a communicator
    if ok:
        return world
    msg = err if err is not None else "another rank could not join the RCCL communicators"
    if required:
        raise L.VaeAmdError(f"vae_amd.parallel.native_init (rank {rank}): {msg}")
    import sys
    print(f"vae_amd.parallel (rank {rank}): native RCCL path unavailable ({msg}); every rank uses the "
          "torch.distributed hook", file=sys.stderr)
    return 0


class DataParallelTrainStep(FusedTrainStep):
    """FusedTrainStep on this rank's shard with synchronised BatchNorm and a gradient all-reduce.

    `B` is the per-rank batch.  The step is enqueued on the current stream (native path: optionally captured,
    collectives included): randn, forward (+ one all-gather of 4*C floats per BatchNorm), ELBO, backward
    (+ one all-gather per BatchNorm), all-reduce SUM of the flat gradient buffer, Adam.  The gradient goes in two
    buckets: backward runs the decoder's ops first (vae_run_backward_range), the all-reduce of the decoder's
    slice of the flat buffer starts on RCCL's stream, and the encoder's backward runs meanwhile; the encoder's
    bucket follows.  `loss` is this rank's shard loss; `global_loss()` sums it over the ranks."""

    def __init__(self, model, optimizer, B, device, bce=None, group=None, native=None, use_graph=False,
                 single_comm=None, allow_seq1d=True):
        """native: the exchanges are enqueued by the library itself over RCCL (vae_dp_*: no Python between the kernels
        of a step; default on the nccl backend); otherwise through torch.distributed from the SyncBN callback (gloo:
        CPU tests, ranks sharing one GPU).  use_graph (native only): the whole step, collectives included, is captured
        into one hipGraph and replayed.  single_comm (native only; None: environment variable VAE_DP_SINGLE_COMM):
        one RCCL communicator and ONE stream for the SyncBN gathers and the gradient all-reduces -- no two collectives
        in flight together, no overlap, the gradient in one all-reduce behind the backward pass; what bench.py runs by
        default at N > 1, and the fallback should the two-communicator overlap stall on a system."""
        # The 1-D families keep their whole-model path (one launch per BatchNorm-delimited phase): between two phases
        # every rank folds its workgroups' statistics rows into one row, the rows are all-gathered through the same
        # hook, and the next phase's prologue merges `world` rows (seq1d_host.h: sq_exchange).  allow_seq1d=False:
        # the per-layer program.
        super().__init__(model, optimizer, B, device, bce=bce, use_graph=False, allow_seq1d=allow_seq1d)
        self.coll = Collective(group)
        self.native = (self.coll.backend == "nccl" and self.device.type == "cuda") if native is None else bool(native)
        self.hook_calls = 0  # SyncBN callbacks into Python (0 on the native path)
        if self.native and native is None:
            # auto mode: a librccl that some rank cannot open or join sends EVERY rank (native_init agrees on the
            # outcome collectively) back to the torch.distributed callback path instead of ending the job
            self.native = native_init(group, self.device, required=False, single_comm=single_comm) > 0
        elif self.native:
            native_init(group, self.device, required=True, single_comm=single_comm)
        if self.native:
            # single-communicator mode: every collective of the step goes to the compute stream (no side stream)
            self.single_comm = bool(L.lib().vae_dp_single_comm())
            self.comm_stream = None if self.single_comm else torch.cuda.Stream(device=self.device)
            self.use_graph = bool(use_graph)
        elif use_graph:
            raise ValueError("use_graph needs the native RCCL path")
        if self.coll.world > L.MAX_WORLD:
            raise ValueError(f"world size {self.coll.world} > {L.MAX_WORLD}")
        # parameters must start identical: rank 0's copy wins (state too)
        for t in (self.flat, self.sflat):
            if t.numel():
                self._broadcast(t)
        # a different eps stream per rank, in a state of this stepper's own: the model's shared Philox state is left
        # alone (XOR-ing it in place cancelled out when a second stepper was built on the same model)
        self.rng = self.rng.clone()
        self.rng[0] = (int(self.rng[0].item()) + (self.coll.rank + 1) * 0x9E3779B1) & ((1 << 62) - 1)
        self._hook_error = None
        self._hook = L.ALLGATHER_FN(self._allgather_hook)  # keep the thunk alive
        self.buckets = self.prog.grad_buckets()  # None: parameters interleave, one bucket

    def _broadcast(self, t):
        if self.coll._staged(t):
            h = t.cpu()
            dist.broadcast(h, src=0, group=self.coll.group)
            t.copy_(h)
        else:
            dist.broadcast(t, src=0, group=self.coll.group)

    def _allgather_hook(self, _user, buf, n, _stream):
        self.hook_calls += 1
        try:
            off = (buf - self.work.data_ptr()) // 4
            if off < 0 or off + self.coll.world * n > self.work.numel():
                raise RuntimeError("all-gather hook: buffer outside the workspace")
            self.coll.all_gather_rows(self.work[off:off + self.coll.world * n].view(self.coll.world, n))
            return 0
        except Exception as e:  # exceptions cannot cross the C frame: hand it to step()
            self._hook_error = e
            return 1

    def _first_eps(self):
        pass  # (this stepper draws its noise at the head of every step, from its own per-rank state)

    def _enqueue_native(self, stream):
        """Everything is enqueued from here in stream order: kernels and SyncBN all-gathers on `stream`, the gradient
        buckets' all-reduces on the communication stream between two events each, Adam behind the last of them."""
        lib = L.lib()
        raw = stream.value if isinstance(stream, ctypes.c_void_p) else stream
        cs = torch.cuda.ExternalStream(raw, device=self.device) if raw else torch.cuda.current_stream(self.device)
        if not self.inject_eps:
            L.check(lib.vae_randn(L.ptr(self.eps), self.eps.numel(), L.ptr(self.rng), stream), "vae_randn")
        ext = [self.x.view(self.B, -1), self.eps]
        L.check(lib.vae_dp_enable(1), "vae_dp_enable")
        try:
            self.prog.forward(self.work, self.flat, self.sflat, ext, True, stream=stream)
            buckets = self.buckets if self.buckets is not None else [(0, self.prog.n_ops, 0, None)]
            if self.comm_stream is None:
                # one communicator, one stream: nothing can overlap, so the whole gradient goes in ONE all-reduce behind
                # the backward pass (two buckets in stream order would expose two collective latencies instead of one)
                buckets = [(0, self.prog.n_ops, 0, None)]
            done = []
            for i, (begin, end, lo, hi) in enumerate(buckets):
                self.prog.backward_range(self.work, self.flat, self.grads, self.sflat, ext, begin, end, i == 0,
                                         stream=stream)
                g = self.grads[lo:hi]
                if self.comm_stream is None:  # single communicator: in stream order on the compute stream
                    L.check(lib.vae_dp_all_reduce_sum(L.ptr(g), g.numel(), stream), "vae_dp_all_reduce_sum")
                    continue
                ready = torch.cuda.Event()
                ready.record(cs)
                self.comm_stream.wait_event(ready)
                L.check(lib.vae_dp_all_reduce_sum(L.ptr(g), g.numel(), ctypes.c_void_p(self.comm_stream.cuda_stream)),
                        "vae_dp_all_reduce_sum")
                ev = torch.cuda.Event()
                ev.record(self.comm_stream)
                done.append(ev)
            for ev in done:
                cs.wait_event(ev)
        finally:
            lib.vae_dp_enable(0)
        self.opt.launch(self.group, self.flat, self.grads, self.m, self.v, self.tick, stream=stream, hyper=self.hyper)

    def _enqueue(self, stream):
        if self.native:
            return self._enqueue_native(stream)
        lib = L.lib()
        if not self.inject_eps:
            L.check(lib.vae_randn(L.ptr(self.eps), self.eps.numel(), L.ptr(self.rng), stream), "vae_randn")
        ext = [self.x.view(self.B, -1), self.eps]
        L.check(lib.vae_set_collective(self._hook, None, self.coll.rank, self.coll.world), "vae_set_collective")
        try:
            self.prog.forward(self.work, self.flat, self.sflat, ext, True, stream=stream)
            pending = []
            if self.buckets is None:
                self.prog.backward(self.work, self.flat, self.grads, self.sflat, ext, stream=stream)
                pending.append(self.coll.all_reduce_sum_async(self.grads))
            else:
                for i, (begin, end, lo, hi) in enumerate(self.buckets):
                    self.prog.backward_range(self.work, self.flat, self.grads, self.sflat, ext, begin, end, i == 0,
                                             stream=stream)
                    pending.append(self.coll.all_reduce_sum_async(self.grads[lo:hi]))
        except L.VaeAmdError:
            if self._hook_error is not None:
                e, self._hook_error = self._hook_error, None
                raise e
            raise
        finally:
            lib.vae_set_collective(ctypes.cast(None, L.ALLGATHER_FN), None, 0, 1)
        for h in pending:
            if h is not None:
                h.wait()
        self.opt.launch(self.group, self.flat, self.grads, self.m, self.v, self.tick, stream=stream, hyper=self.hyper)

    def global_loss(self):
        """Sum of the shard losses of the last step (a host float; synchronises)."""
        t = self.loss.detach().clone()
        self.coll.all_reduce_sum(t)
        return float(t.item())
