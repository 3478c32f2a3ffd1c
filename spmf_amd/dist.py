"""Row-sharded data parallelism for the hot path.

The energy's data term and z-prior are sums over rows (poisson.py:604,
617-618); priors depend only on replicated parameters.  So: contiguous row
shards of the count matrix, one process per GPU, and ONE sum all-reduce per
step over the packed fp32 accumulators the data pass leaves in the workspace
(``[gA' | gV' | gphi | fp64 scalars as (hi,lo) pairs]``), after which every
rank runs the finish kernel redundantly.  The reference has no counterpart
(only the ``strategy`` pass-through, poisson.py:60,72).

``torch.distributed`` is transport only: backend "nccl" is RCCL over xGMI on
the GPU box, "gloo" drives the CPU tests.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(total_rows: int, world: int, rank: int, granule: int = 1, row_ptr=None):
    """Contiguous [r0, r1) of this rank; shard edges fall on multiples of
    ``granule`` rows (generator chunks / panels) and cover every row once.

    Without ``row_ptr`` the shards hold equal counts of granules.  With the CSR
    row-pointer array of the whole matrix (``row_ptr[total_rows]`` = nnz; numpy,
    torch or a list) the edges are the granule boundaries nearest to equal STORED
    counts (SURVEY 8e "nnz-balanced split points"): a step's time follows the
    stored entries of the shard, so deep / shallow row blocks no longer make one
    rank the slowest.  Edges stay strictly increasing while there are at least
    ``world`` granules, so no rank is left without rows."""
    units = -(-total_rows // granule)
    if row_ptr is None:
        u0 = units * rank // world
        u1 = units * (rank + 1) // world
        return min(total_rows, u0 * granule), min(total_rows, u1 * granule)
    import numpy as np
    rp = np.asarray(row_ptr.cpu() if hasattr(row_ptr, "cpu") else row_ptr, dtype=np.int64)
    if rp.shape[0] != total_rows + 1:
        raise ValueError("row_ptr must have total_rows + 1 entries")
    edges_rows = np.minimum(np.arange(units + 1, dtype=np.int64) * granule, total_rows)
    cum = rp[edges_rows] - rp[0]                  # stored entries before each granule boundary
    nnz = int(cum[-1])
    cuts = [0]
    for r in range(1, world):
        target = nnz * r / world
        j = int(np.searchsorted(cum, target))     # first boundary with cum >= target
        if j > 0 and (j > units or target - cum[j - 1] <= cum[min(j, units)] - target):
            j -= 1
        lo = cuts[-1] + 1 if units >= world else cuts[-1]
        hi = units - (world - r) if units >= world else units
        cuts.append(int(min(max(j, lo), max(hi, lo))))
    cuts.append(units)
    return int(edges_rows[min(cuts[rank], units)]), int(edges_rows[min(cuts[rank + 1], units)])


class LibraryComm:
    """The library's own RCCL communicator (include/spmf_hip.h: spmf_comm_init /
    spmf_allreduce): ncclAllReduce issued by libspmf_hip.so on the caller's stream,
    so the step's collective is stream-ordered with the kernels around it and torch
    is not on the data path.  torch.distributed (any backend) only carries the 128-byte
    unique id from rank 0 to the others; with one rank nothing travels at all."""

    def __init__(self, model, rank=None, world=None, group=None):
        import ctypes as C
        from . import _lib
        self.model = model
        lib, h = _lib.load(), model._handle()
        if rank is None:
            rank = dist.get_rank(group) if dist.is_initialized() else 0
        if world is None:
            world = dist.get_world_size(group) if dist.is_initialized() else 1
        ident = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            buf = (C.c_char * 128)()
            rc = lib.spmf_comm_unique_id(buf)
            if rc != 0:
                raise _lib.SpmfError(f"spmf_comm_unique_id failed (rc={rc}): librccl not loadable")
            ident = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
        if world > 1:
            if dist.get_backend(group) == "nccl":
                dev = ident.to(model.device)
                dist.broadcast(dev, 0, group=group)
                ident = dev.cpu()
            else:
                dist.broadcast(ident, 0, group=group)
        raw = (C.c_char * 128).from_buffer_copy(bytes(ident.numpy().tobytes()))
        _lib.check(h, lib.spmf_comm_init(h, raw, int(rank), int(world)), "spmf_comm_init")
        self.rank, self.world = int(rank), int(world)

    def all_reduce_(self, t):
        from . import _lib
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise ValueError("LibraryComm reduces contiguous fp32 device tensors (the packed accumulators)")
        lib, h = _lib.load(), self.model._handle()
        stream = torch.cuda.current_stream(t.device).cuda_stream
        _lib.check(h, lib.spmf_allreduce(h, t.data_ptr(), t.numel(), stream), "spmf_allreduce")


class PeerComm:
    """The step's collective as the library's own kernel over peer pointers
    (include/spmf_hip.h spmf_p2p_*; csrc/p2p.hip): a direct reduce-scatter + all-gather in
    which every rank pushes its slices straight into the owners' inboxes -- over xGMI when the
    ranks are GPUs of one node, through same-device IPC mappings when they are processes on
    one GPU (how the one-GPU tests run it at world 2 and 4).  One kernel launch per rank on
    the caller's stream: stream-ordered, capturable in a hipGraph (the call counter is on the
    device), and every rank ends with the same bits (each slice is reduced once, by its owner,
    in rank order).  torch.distributed (any backend) only carries the 64-byte IPC handles at
    construction.  ``n_max``: the longest buffer that will be reduced (default: the packed
    accumulators of ``max_draws`` draws)."""

    def __init__(self, model, n_max=None, max_draws=1, nchunk=0, rank=None, world=None, group=None):
        import ctypes as C
        from . import _lib
        self.model = model
        lib, h = _lib.load(), model._handle()
        if rank is None:
            rank = dist.get_rank(group) if dist.is_initialized() else 0
        if world is None:
            world = dist.get_world_size(group) if dist.is_initialized() else 1
        if n_max is None:
            n_max = int(lib.spmf_acc_len(h, int(max_draws)))
        n_max = max(int(n_max), 4 * int(world))
        hnd = (C.c_char * 64)()

        def agree(rc, what):
            """Every rank learns whether ANY rank failed a local step, before the next collective: a rank
            that raised alone would leave the others waiting in it."""
            bad = 1 if rc != 0 else 0
            msg = lib.spmf_last_error(h).decode() if rc != 0 else ""
            if world > 1:
                t = torch.tensor([bad], dtype=torch.int32)
                if dist.get_backend(group) == "nccl":
                    t = t.to(model.device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
                bad_any = int(t.item())
            else:
                bad_any = bad
            if bad_any:
                lib.spmf_p2p_destroy(h)
                raise _lib.SpmfError(f"PeerComm: {what} failed on " +
                                     (f"this rank (rc={rc}): {msg}" if bad else "another rank"))
        agree(lib.spmf_p2p_init(h, int(rank), int(world), n_max, int(nchunk), hnd), "spmf_p2p_init")
        mine = torch.frombuffer(bytearray(hnd.raw), dtype=torch.uint8).clone()
        if world > 1:
            if dist.get_backend(group) == "nccl":
                mine = mine.to(model.device)
            every = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(every, mine, group=group)
            blob = b"".join(bytes(t.cpu().numpy().tobytes()) for t in every)
        else:
            blob = bytes(mine.numpy().tobytes())
        raw = (C.c_char * (64 * world)).from_buffer_copy(blob)
        agree(lib.spmf_p2p_connect(h, raw), "spmf_p2p_connect")
        if world > 1:
            dist.barrier(group=group)      # every rank has mapped every region before the first push
        self.rank, self.world, self.n_max = int(rank), int(world), n_max
        self.kind = "p2p"

    def all_reduce_(self, t):
        from . import _lib
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise ValueError("PeerComm reduces contiguous fp32 device tensors (the packed accumulators)")
        lib, h = _lib.load(), self.model._handle()
        stream = torch.cuda.current_stream(t.device).cuda_stream
        _lib.check(h, lib.spmf_allreduce(h, t.data_ptr(), t.numel(), stream), "spmf_allreduce")

    def enable(self, on=True):
        """Which transport spmf_allreduce uses on this model's context when it also holds an RCCL
        communicator (LibraryComm): this kernel (True) or RCCL (False)."""
        from . import _lib
        lib, h = _lib.load(), self.model._handle()
        _lib.check(h, lib.spmf_p2p_enable(h, 1 if on else 0), "spmf_p2p_enable")

    def status(self):
        """(calls completed, first call in which a workgroup gave up waiting for a peer or 0).
        Synchronises the device."""
        import ctypes as C
        from . import _lib
        lib, h = _lib.load(), self.model._handle()
        out = (C.c_uint64 * 3)()
        _lib.check(h, lib.spmf_p2p_status(h, out), "spmf_p2p_status")
        return int(out[0]), int(out[2])

    def close(self, group=None, sync=True):
        """Orderly shutdown (collective when torch.distributed is up and ``sync``): every rank unmaps its peers'
        regions and the ranks meet -- so no region is freed (with its model's context) while a peer still has it
        mapped.  ``sync=False``: this rank alone (error paths)."""
        from . import _lib
        lib, h = _lib.load(), self.model._handle()
        _lib.check(h, lib.spmf_p2p_disconnect(h), "spmf_p2p_disconnect")
        if sync and self.world > 1 and dist.is_initialized():
            dist.barrier(group=group)
        # (the own region stays with the model's context -- a later PeerComm on the same model re-uses it, and it
        #  is freed with the context: regions are allocated once per context, spmf_p2p_init)


class ShardReducer:
    """Holds the global batch constants and performs the per-step all-reduce.
    ``comm``: a LibraryComm moves the packed fp32 accumulators with the library's
    own RCCL communicator; without one (default) torch.distributed does."""

    def __init__(self, group=None, comm=None, overlap_prior=None, capture=None):
        self.group = group
        self.comm = comm
        self.capture = capture         # None: see graph_safe
        self.active = dist.is_initialized() or comm is not None   # also with one rank (exercises the transport)
        self.world = dist.get_world_size(group) if dist.is_initialized() else (comm.world if comm else 1)
        self.rank = dist.get_rank(group) if dist.is_initialized() else (comm.rank if comm else 0)
        self.rows_global = None        # per-step batch totals when every batch is the whole shard
        self.lgamma_global = None
        self.dataset_rows = None       # set by reduce_stats
        self.dataset_lgamma = None
        # the prior half of the finish on the library's side stream, under the collective
        # (spmf_prior_async): worth its fork/join only when the collective takes time, i.e.
        # with more than one rank (one rank, measured: the fork/join costs what the 22 us
        # prior half saves)
        self.overlap_prior = (self.world > 1) if overlap_prior is None else bool(overlap_prior)
        self._totals_cache = {}        # id(batch struct) -> (struct, (rows, lgamma)) for batch_totals

    @property
    def graph_safe(self):
        """The step's collective may be captured in a hipGraph: spmf_allreduce is a plain
        stream-ordered ncclAllReduce on the capturing stream.  torch.distributed's own
        collectives (and the host-staged gloo rehearsal) stay eager.  The capture has only
        ever run with ONE rank (no multi-GPU box so far), so with more ranks it is opt-in
        (``capture=True``): the default there is the same device-gated step as plain launches,
        which costs a few percent on launch-bound batches and nothing on GPU-bound ones
        (profiles/r04_graph_threshold_sweep.txt)."""
        if self.capture is not None:
            return bool(self.capture) and (self.comm is not None or not self.active)
        if getattr(self.comm, "kind", None) == "p2p":
            return True    # a kernel launch whose call counter lives on the device (tests: world 2 and 4 replayed)
        return (self.comm is not None and self.world == 1) or not self.active

    def sum_(self, acc):
        """The step's one collective, nothing else (no totals, no host read)."""
        if self.active:
            self._sum(acc)

    def batch_totals(self, cs):
        """Global (rows, lgamma) of the batch this rank's struct `cs` belongs to, reduced ONCE
        per batch object and remembered: the device-resident training loop (vi.vi_step_dev)
        asks every step and must not read anything back.  Valid while rank r's i-th batch
        always meets the same batches of the other ranks (a fixed data factory, the same
        number of batches per epoch on every rank: a rank that misses where the others hit
        would enter the collective alone).  A miss inside a hipGraph capture raises."""
        if self.rows_global is not None:
            return self.rows_global, self.lgamma_global
        hit = self._totals_cache.get(id(cs))
        if hit is None or hit[0] is not cs:
            if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("ShardReducer.batch_totals: the totals of this batch are not known yet and a "
                                   "hipGraph capture is in progress (their reduction is a collective and a host "
                                   "read): run the step eagerly once before capturing it")
            if len(self._totals_cache) > 4096:
                self._totals_cache.clear()
            hit = (cs, self.totals(cs.n_rows, cs.lgamma_sum))
            self._totals_cache[id(cs)] = hit
        return hit[1]

    def share_int(self, value):
        """Rank 0's non-negative integer (< 2**63) on every rank, moved in float32-exact
        21-bit pieces through gather_scalar (any transport that can sum)."""
        value = int(value)
        out = 0
        for i in range(3):
            piece = (value >> (21 * i)) & ((1 << 21) - 1)
            out |= int(self.gather_scalar(float(piece))[0]) << (21 * i)
        return out

    def reduce_stats(self, colsum, colnnz, rows, lgamma_sum, full_batch=False):
        """compute_scales' one-time reduction (poisson.py:118-135 across
        shards): sums colsum/colnnz in place, returns the dataset's global
        (rows, lgamma) and keeps them as dataset_rows / dataset_lgamma.
        ``full_batch``: every step's batch IS the rank's whole shard, so these are
        also the per-step batch totals (set_batch_totals) and no step has to
        reduce them again; leave it False when the shards are minibatched."""
        tot = torch.tensor([float(rows), float(lgamma_sum)], dtype=torch.float64,
                           device=colsum.device)
        if self.active:
            for t in (colsum, colnnz, tot):
                self._sum(t)
        self.dataset_rows = int(round(float(tot[0])))
        self.dataset_lgamma = float(tot[1])
        if full_batch:
            self.set_batch_totals(self.dataset_rows, self.dataset_lgamma)
        return self.dataset_rows, self.dataset_lgamma

    def set_batch_totals(self, rows_global, lgamma_global):
        self.rows_global, self.lgamma_global = int(rows_global), float(lgamma_global)

    def _sum(self, t):
        if self.comm is not None and t.is_cuda and t.dtype == torch.float32:
            self.comm.all_reduce_(t)
            return
        if not dist.is_initialized():
            return                               # one rank, library comm only: nothing else to sum
        if dist.get_backend(self.group) != "nccl" and t.is_cuda:
            h = t.cpu()                      # rehearsal transport (gloo): host-staged
            dist.all_reduce(h, group=self.group)
            t.copy_(h)
        else:
            dist.all_reduce(t, group=self.group)

    # ---- column-split step (PoissonFactorization.energy_and_grads) -----------
    def start(self, piece):
        """Begin the sum all-reduce of one contiguous accumulator range; returns a
        handle for wait().  With RCCL the collective runs on the process group's
        stream, ordered after what is already queued on the current stream, so the
        kernels launched next (the upper half's column pass) overlap it."""
        if not self.active:
            return None
        if self.comm is not None:
            self._sum(piece)                      # stream-ordered on the caller's stream
            return None
        if dist.get_backend(self.group) == "nccl":
            return dist.all_reduce(piece, group=self.group, async_op=True)
        self._sum(piece)                      # rehearsal transport: synchronous
        return None

    def wait(self, work):
        if work is not None:
            work.wait()                       # the current stream waits, not the host

    def _nccl(self):
        """torch.distributed is up AND its backend is RCCL (device tensors only)."""
        return dist.is_initialized() and dist.get_backend(self.group) == "nccl"

    def totals(self, rows, lgamma_sum):
        """Global (rows, lgamma) of the batch, as __call__ returns them."""
        if self.rows_global is None:
            dev = torch.device("cuda", torch.cuda.current_device()) if self._nccl() \
                else torch.device("cpu")
            tot = torch.tensor([float(rows), float(lgamma_sum)], dtype=torch.float64, device=dev)
            if self.active:
                self._sum(tot)       # fp64: torch.distributed's job; a no-op with a library comm alone
            return int(round(float(tot[0]))), float(tot[1])
        return self.rows_global, self.lgamma_global

    def gather_scalar(self, value, device=None):
        """[world] list of every rank's float32-exact scalar (the non-finite rule's
        per-shard minimum, poisson.py:609): a one-hot vector through the SAME sum
        all-reduce as the accumulators, so any transport that can sum can gather."""
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if (
                self.comm is not None or (dist.is_initialized() and dist.get_backend(self.group) == "nccl")
            ) else torch.device("cpu")
        v = torch.zeros(self.world, dtype=torch.float32, device=device)
        v[self.rank] = float(value)
        if self.active:
            self._sum(v)
        return [float(x) for x in v.cpu()]

    # ---- replicated state ------------------------------------------------------
    def sync_replicas(self, tensors, src=0):
        """Broadcast rank `src`'s copy of the replicated tensors (surrogate
        trainables, Adam moments).  Every rank applies the same update to the same
        all-reduced accumulators and the finish kernel sums across blocks in a fixed
        order, so the replicas are expected to stay bit-identical; the driver still
        re-broadcasts every few hundred steps (one small broadcast) as a guard against
        anything rank dependent upstream (replicas_max_abs_diff measures it).  Needs a
        torch.distributed process group: with one rank, or with a library communicator
        alone, there is nothing to broadcast."""
        if not self.active or self.world == 1 or not dist.is_initialized():
            return
        for t in tensors:
            if dist.get_backend(self.group) != "nccl" and t.is_cuda:
                h = t.detach().cpu()
                dist.broadcast(h, src, group=self.group)
                t.detach().copy_(h)
            else:
                dist.broadcast(t.detach(), src, group=self.group)

    def replicas_max_abs_diff(self, tensors):
        """max over ranks and tensors of |t - rank 0's t| (a diagnostic: 0.0 means
        the replicas are bit-identical)."""
        worst = 0.0
        for t in tensors:
            ref = t.detach().clone()
            self.sync_replicas([ref])
            d = (t.detach() - ref).abs().max().reshape(1).double() if t.numel() \
                else torch.zeros(1, dtype=torch.float64, device=t.device)
            if dist.is_initialized() and self.world > 1:
                if not self._nccl():
                    d = d.cpu()                  # gloo reduces host tensors, RCCL device tensors
                dist.all_reduce(d, op=dist.ReduceOp.MAX, group=self.group)
            worst = max(worst, float(d))
        return worst

    def __call__(self, acc, rows, lgamma_sum):
        """all_reduce hook of PoissonFactorization.energy_and_grads."""
        if self.active:
            self._sum(acc)
        if self.rows_global is None:
            tot = torch.tensor([float(rows), float(lgamma_sum)], dtype=torch.float64,
                               device=acc.device)
            if self.active:
                self._sum(tot)
            return int(round(float(tot[0]))), float(tot[1])
        return self.rows_global, self.lgamma_global


def make_reducer(model, prefer="auto", group=None, max_draws=1):
    """A ShardReducer over the best transport this job has: the library's peer-pointer kernel (PeerComm:
    stream-ordered, hipGraph-capturable at any world size, the same bits on every rank), else its RCCL
    communicator (LibraryComm; torch.distributed backend "nccl" only), else torch.distributed itself.
    ``prefer``: "auto" | "p2p" | "rccl" | "torch".  Every rank must call it (the set-up is collective and a
    failure on one rank is agreed on by all).  bench.py additionally TIMES the first two against each other."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1 or model.device.type != "cuda" \
            or prefer == "torch":
        return ShardReducer(group=group)
    comm = None
    if prefer in ("auto", "p2p"):
        try:
            comm = PeerComm(model, max_draws=max_draws, group=group)
        except Exception as e:              # (agreed on by every rank inside PeerComm)
            if prefer == "p2p":
                raise
            print(f"make_reducer: peer-pointer collective unavailable ({e}); trying RCCL")
    if comm is None and dist.get_backend(group) == "nccl":
        try:
            comm = LibraryComm(model, group=group)
        except Exception as e:
            if prefer == "rccl":
                raise
            print(f"make_reducer: library RCCL communicator unavailable ({e}); torch.distributed carries the step")
    return ShardReducer(group=group, comm=comm)


def sync_seed(seed=None, group=None):
    """Every rank must draw the SAME base noise for the replicated surrogate
    (parameters, and hence theta, are replicated; only the rows are sharded).
    Broadcast rank 0's seed and seed torch's generators with it."""
    import random
    if seed is None:
        seed = random.SystemRandom().randrange(2 ** 31)
    t = torch.tensor([int(seed)], dtype=torch.int64)
    if dist.is_initialized():
        dev = torch.device("cuda", torch.cuda.current_device()) \
            if dist.get_backend(group) == "nccl" else torch.device("cpu")
        t = t.to(dev)
        dist.broadcast(t, 0, group=group)
    seed = int(t.item())
    torch.manual_seed(seed)
    return seed
