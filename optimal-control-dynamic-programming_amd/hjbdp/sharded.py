"""Multi-GPU backward sweep: the state grid is block-partitioned along its LAST
(slowest-varying, outermost) axis, one slab per rank, one process per GPU.

Within a stage every backup is independent; J_{k+1} is read at x_next, which for
the spacecraft models lies within a few cells of x, so each rank needs only
`halo_lo`/`halo_hi` neighbouring planes of J_{k+1}.  Per stage (overlap=True, GPUs):

    1. the neighbour halo exchange of the boundary planes of J_{k+1} is STARTED
       (torch.distributed P2P: RCCL send/recv over xGMI; its own stream);
    2. the fused backup kernel runs on the INTERIOR planes - those whose next states
       stay inside the owned planes - while the halos are in flight;
    3. when the halos have landed, the kernel runs on the two boundary strips.

Interior and strips are three libhjbdp slab handles over the SAME J buffers (a slab
handle sees the planes [begin - halo_lo, end + halo_hi) of the last axis; sub-slabs of
a rank's slab are pointer offsets into its buffers), so the split costs no copies.
They live inside the library (hjb_rank_create), and a whole stage - fork, interior,
strips behind the halos on streams of their own, join - is ONE call (hjb_rank_stage):
per stage this module issues the exchange and that call, nothing else.
overlap=False (and every CPU test) exchanges first and runs one kernel on all owned planes.

The reference has no distributed code at all; this is new design
for the sweep loops of Solver_position.m:132-141 / Solver_pos_att.m:270-286.
The early-stop monitor's two sums (Solver_pos_att.m:273-285) become one
all-reduce per monitor point.

`stage_fn` is the per-slab backup.  The product default is the HIP library (it
raises if the library or a GPU is missing - no fallback); CPU tests inject the
oracle to exercise partitioning + exchange under gloo.
"""
from __future__ import annotations

import numpy as np

from .problem import ProblemSpec


def partition(n_planes, world):
    """Contiguous, balanced plane ranges [begin, end) per rank."""
    base, rem = divmod(n_planes, world)
    out, b = [], 0
    for r in range(world):
        e = b + base + (1 if r < rem else 0)
        out.append((b, e))
        b = e
    return out


def required_halo(spec: ProblemSpec):
    """Conservative (lo, hi) plane counts a slab needs on each side, from the
    stage-invariant tables of the last axis: for every plane i the range of
    x_next over all other dims/controls is bounded by the sum of per-term extrema.
    Mirrors the computation hjb_create reports in hjb_info.halo_needed_*."""
    a = spec.D - 1
    n = spec.n[a]
    lo = np.zeros(n)
    hi = np.zeros(n)
    for t in spec.next_terms[a]:
        d = t.data.astype(np.float64)
        if a in t.dims:
            ax = t.dims.index(a)
            other = tuple(i for i in range(d.ndim) if i != ax)
            lo += d.min(axis=other) if other else d
            hi += d.max(axis=other) if other else d
        else:
            lo += d.min()
            hi += d.max()
    k = spec.knots[a].astype(spec.dtype)
    span = np.abs(lo) + np.abs(hi)

    def cell(q):
        c = np.searchsorted(k, q.astype(spec.dtype), side="right") - 1
        return np.clip(c, 0, n - 2)
    i = np.arange(n)
    need_lo = int(np.max(i - cell(lo - 1e-6 * span)))
    need_hi = int(np.max(cell(hi + 1e-6 * span) + 1 - i))
    return max(need_lo, 0), max(need_hi, 0)


class ShardedSweep:
    """One rank's share of a sweep.  Buffers are torch tensors (HBM on GPUs)."""

    def __init__(self, spec: ProblemSpec, rank, world, device, stage_fn=None, group=None, overlap=False, transport="torch"):
        """transport: "torch" - the halo planes move with torch.distributed P2P (RCCL on GPUs, gloo in the CPU / shared-GPU
        tests), issued from Python every stage; "lib" - with the RCCL transport INSIDE libhjbdp (hjb_rank_comm_init /
        hjb_rank_step: one library call per stage; torch.distributed only carries the 128-byte communicator id once)."""
        import torch
        self.torch = torch
        self.transport = transport
        self.spec, self.rank, self.world, self.group = spec, int(rank), int(world), group
        self.device = torch.device(device)
        nl = spec.n[-1]
        if world > nl:
            raise ValueError("more ranks than planes")
        self.parts = partition(nl, world)
        self.begin, self.end = self.parts[self.rank]
        need_lo, need_hi = required_halo(spec)
        self.halo_lo = min(need_lo, self.begin)
        self.halo_hi = min(need_hi, nl - self.end)
        for r, (b, e) in enumerate(self.parts):
            # a halo must come from the immediate neighbour only
            if r > 0 and min(need_lo, b) > self.parts[r - 1][1] - self.parts[r - 1][0]:
                raise ValueError("halo wider than the neighbouring slab: use fewer ranks")
            if r < world - 1 and min(need_hi, nl - e) > self.parts[r + 1][1] - self.parts[r + 1][0]:
                raise ValueError("halo wider than the neighbouring slab: use fewer ranks")
        self.inner = spec.nS // nl
        self.owned = self.end - self.begin
        self.nplanes = self.owned + self.halo_lo + self.halo_hi
        tdt = {np.dtype(np.float16): torch.float16, np.dtype(np.float32): torch.float32,
               np.dtype(np.float64): torch.float64}[np.dtype(spec.j_dtype)]
        # haloed J layout [plane, inner] (row-major torch view of column-major [inner, plane])
        self.J = [torch.zeros((self.nplanes, self.inner), dtype=tdt, device=self.device) for _ in range(2)]
        # argmin labels in the width the problem asks for (hjb_problem.idx_dtype): int32, uint8 or uint16
        idt = {4: torch.int32, 1: torch.uint8, 2: torch.uint16}[spec.idx_np_dtype.itemsize]
        self.idx = torch.zeros((self.owned, self.inner), dtype=idt, device=self.device)
        self.cur = 0
        self.slab = (self.begin, self.end, self.halo_lo, self.halo_hi)
        self._rank = None             # the HIP path: the library's view of this rank (hjb_rank_create)
        if stage_fn is None:
            from .core import RankSlab    # raises loudly without the library / a GPU
            dev_index = self.device.index if self.device.index is not None else 0
            self._rank = RankSlab(spec, dev_index, self.rank, world, overlap=overlap)
            rk = self._rank
            if (rk.begin, rk.end, rk.halo_lo, rk.halo_hi) != self.slab:
                raise RuntimeError("library partition %r differs from the host's %r" % ((rk.begin, rk.end, rk.halo_lo, rk.halo_hi), self.slab))
            stage_fn = self._hip_stage
            if rk.split:
                self._comm_stream = torch.cuda.Stream(device=self.device)
            if transport == "lib" and world > 1:
                import ctypes as C
                import torch.distributed as dist
                uid = (C.c_char * 128)()
                # Before the collective ncclCommInitRank: can EVERY rank's library reach RCCL at all?  (hjb_rank_comm_available:
                # dlopen + symbols, nothing else.)  Agreed on through the process group, so that a rank that cannot raises on ALL
                # ranks - a caller that tries this transport beside another one (bench.py) can catch it without leaving the
                # other ranks waiting inside RCCL.
                ok = 1 if rk.lib.hjb_rank_comm_available() == 0 else 0
                why = "" if ok else ((rk.lib.hjb_rank_last_error(None) or b"").decode() or "status != HJB_OK")
                flag = torch.tensor([ok], dtype=torch.int32, device=self.device if dist.get_backend(group) == "nccl" else "cpu")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
                if int(flag.item()) == 0:
                    raise RuntimeError("transport 'lib': the RCCL transport inside libhjbdp is unavailable on at least one rank"
                                       + (" (this rank: %s)" % why if why else ""))
                if self.rank == 0:
                    rk._check(rk.lib.hjb_rank_comm_unique_id(uid))
                box = [bytes(uid)]
                dist.broadcast_object_list(box, src=0, group=group)
                rk._check(rk.lib.hjb_rank_comm_init(rk._r, box[0]))
        self.stage_fn = stage_fn
        self._halo_ops = {}
        self.post_exchange = True     # split form: strips first, the exchange of the output behind them (False: exchange the input first)
        self._halos_valid = False     # the current buffer's halo planes hold the neighbours' boundary planes
        # what my neighbours need from me
        self.up_needs = min(need_lo, self.end) if self.rank < world - 1 else 0     # my top planes -> rank+1's lower halo
        self.dn_needs = min(need_hi, nl - self.begin) if self.rank > 0 else 0      # my bottom planes -> rank-1's upper halo

    def _hip_stage(self, J_in, J_out, idx):
        stream = self.torch.cuda.current_stream(self.device).cuda_stream
        self._rank.stage(J_in, J_out, idx, compute_stream=stream)

    # -- the handle(s) behind this rank ------------------------------------------------------------------
    def info(self):
        rk = self._rank
        return {"kernel_variant": rk.kernel_variant, "halo_needed_lo": rk.need_lo, "halo_needed_hi": rk.need_hi,
                "idx_bytes": rk.idx_bytes, "split": rk.split}

    def comm_ranks(self):
        """The rank count the transport's communicator ITSELF reports: ncclCommCount through the library (transport "lib"),
        the process group's size (transport "torch"); 1 without a communicator."""
        if self.world == 1:
            return 1
        if self.transport == "lib" and self._rank is not None:
            import ctypes as C
            n, me = C.c_int32(-1), C.c_int32(-1)
            self._rank._check(self._rank.lib.hjb_rank_comm_info(self._rank._r, C.byref(n), C.byref(me)))
            if me.value not in (-1, self.rank):
                raise RuntimeError("communicator rank %d != rank %d" % (me.value, self.rank))
            return int(n.value)
        import torch.distributed as dist
        return int(dist.get_world_size(self.group))

    def set_option(self, key, value):
        self._rank.set_option(key, value)

    def get_option(self, key):
        return self._rank.get_option(key)

    def check_device_status(self):
        if getattr(self, "_comm_stream", None) is not None:       # a trailing exchange (post_exchange) may still be in flight
            self.torch.cuda.current_stream(self.device).wait_stream(self._comm_stream)
        if self.transport == "lib" and self.world > 1 and self._rank is not None:
            xs = self._rank.lib.hjb_rank_transfer_stream(self._rank._r)
            if xs:
                self._rank.check_device_status(xs)                 # ... on the library's transfer stream: synchronises it
        self._rank.check_device_status(self.torch.cuda.current_stream(self.device).cuda_stream)

    def set_terminal(self, J_global=None):
        """J_N: None = zeros (Dynamic_Solver.m:83-84); else global [nS] column-major."""
        J = self.J[self.cur]
        J.zero_()
        self._halos_valid = False
        if J_global is not None:
            g = self.torch.as_tensor(np.asarray(J_global, dtype=self.spec.j_dtype).reshape(self.spec.n[-1], self.inner))
            J[self.halo_lo:self.halo_lo + self.owned] = g[self.begin:self.end].to(self.device)

    def exchange_halos(self, wait=True, which=None):
        """Fill the halo planes of the current J (which=None) or of buffer `which` from the neighbouring ranks.  wait=False
        returns the pending work objects instead of waiting (RCCL: the transfers run on the communicator's stream)."""
        if self.world == 1:
            return []
        import torch.distributed as dist
        which = self.cur if which is None else which
        J = self.J[which]
        cached = self._halo_ops.get(which)              # the two J buffers never move: their send / receive views are built once
        if cached is None:
            ops, keep = [], []
            lo0 = self.halo_lo
            if self.rank > 0:
                if self.dn_needs:
                    s = J[lo0:lo0 + self.dn_needs]          # whole planes of a [planes, inner] buffer: contiguous views
                    keep.append(s)
                    ops.append(dist.P2POp(dist.isend, s, self.rank - 1, group=self.group))
                if self.halo_lo:
                    ops.append(dist.P2POp(dist.irecv, J[0:self.halo_lo], self.rank - 1, group=self.group))
            if self.rank < self.world - 1:
                if self.up_needs:
                    s = J[lo0 + self.owned - self.up_needs:lo0 + self.owned]
                    keep.append(s)
                    ops.append(dist.P2POp(dist.isend, s, self.rank + 1, group=self.group))
                if self.halo_hi:
                    ops.append(dist.P2POp(dist.irecv, J[lo0 + self.owned:], self.rank + 1, group=self.group))
            assert all(o.tensor.is_contiguous() for o in ops)
            cached = self._halo_ops[which] = (ops, keep)
        ops, keep = cached
        if ops:
            if J.is_cuda and dist.get_backend(self.group) == "gloo":
                # test-only transport (several ranks sharing one GPU): gloo moves host memory
                staged = []
                for op in ops:
                    host = op.tensor.cpu() if op.op is dist.isend else self.torch.empty(
                        op.tensor.shape, dtype=op.tensor.dtype)
                    staged.append((op, host))
                works = dist.batch_isend_irecv([dist.P2POp(op.op, host, op.peer, group=self.group) for op, host in staged])
                for w in works:
                    w.wait()
                for op, host in staged:
                    if op.op is dist.irecv:
                        op.tensor.copy_(host)
                return []
            works = dist.batch_isend_irecv(ops)
            if not wait:
                self._keep = keep
                return works
            for w in works:
                w.wait()
        return []

    def step(self):
        """One backup of the owned planes.  Without overlap: halo exchange, then the fused kernel.  With
        overlap: start the exchange, run the interior planes, wait for the halos, run the boundary strips."""
        J_in, J_out = self.J[self.cur], self.J[1 - self.cur]
        if self._rank is not None and self.transport == "lib" and self.world > 1:
            rk = self._rank
            stream = self.torch.cuda.current_stream(self.device).cuda_stream
            if self.post_exchange:       # strips first, the exchange of J_out's boundary planes behind them (hjb_rank_step_post)
                if not self._halos_valid:
                    rk._check(rk.lib.hjb_rank_exchange(rk._r, J_in.data_ptr(), stream))
                rk._check(rk.lib.hjb_rank_step_post(rk._r, J_in.data_ptr(), J_out.data_ptr(), self.idx.data_ptr(), stream))
                self._halos_valid = True
            else:
                rk._check(rk.lib.hjb_rank_step(rk._r, J_in.data_ptr(), J_out.data_ptr(), self.idx.data_ptr(), stream))
                self._halos_valid = False                  # J_out's halo planes are NOT exchanged by this order
        elif self._rank is None or not self._rank.split:
            self.exchange_halos()
            self.stage_fn(J_in, J_out, self.idx)
            self._halos_valid = False
        elif self.post_exchange:
            # Boundary strips FIRST (the halos of J_in arrived during the previous stage), the interior beside them, and the
            # exchange of J_OUT's boundary planes as soon as the strips are done - it has the rest of the interior to complete,
            # and the next stage starts with its halos in place (hjb_rank_step_post's order, include/hjbdp.h).
            t = self.torch
            main = t.cuda.current_stream(self.device)
            if not self._halos_valid:                      # the terminal cost: one exchange before the first stage
                self._comm_stream.wait_stream(main)
                with t.cuda.stream(self._comm_stream):
                    for w in self.exchange_halos(wait=False):
                        w.wait()
            self._rank.stage_post(J_in, J_out, self.idx, compute_stream=main.cuda_stream, halo_stream=self._comm_stream.cuda_stream)
            if not self._rank.wait_strips(self._comm_stream.cuda_stream):
                self._comm_stream.wait_stream(main)        # a neighbour needs planes the strips do not cover: after the interior
            with t.cuda.stream(self._comm_stream):
                for w in self.exchange_halos(wait=False, which=1 - self.cur):
                    w.wait()
            self._halos_valid = True
        else:
            t = self.torch
            main = t.cuda.current_stream(self.device)
            self._comm_stream.wait_stream(main)            # the previous stage's output is the data to send
            with t.cuda.stream(self._comm_stream):
                for w in self.exchange_halos(wait=False):
                    w.wait()                               # the transfer stream waits for the transfers
            # interior on the compute stream now, the strips behind an event the library records on the transfer stream
            # at this point (= the halos have landed), on streams of their own; joined into the compute stream
            self._rank.stage(J_in, J_out, self.idx, compute_stream=main.cuda_stream, halo_stream=self._comm_stream.cuda_stream)
            self._halos_valid = False                      # a later switch to post_exchange primes its halos again
        self.cur = 1 - self.cur

    def monitor_sums(self):
        """(sum J, sum idx) over the whole grid: the pos-att monitor's fsum50/idsum50."""
        t = self.torch
        J = self.J[self.cur][self.halo_lo:self.halo_lo + self.owned]
        v = t.stack([J.double().sum(), self.idx.to(t.int32).double().sum()])
        if self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(v, group=self.group)
        return float(v[0]), float(v[1])

    def sweep(self, n_stages, monitor_period=0, monitor_tol=0.0):
        fprev = 0.0
        done = 0
        for k_s in range(n_stages, 0, -1):
            self.step()
            done += 1
            if monitor_period and k_s % monitor_period == 0:
                fsum, _ = self.monitor_sums()
                e, fprev = fsum - fprev, fsum
                if abs(e) < monitor_tol:
                    break
        if self._rank is not None:
            self.check_device_status()
        return done

    def owned_J(self):
        return self.J[self.cur][self.halo_lo:self.halo_lo + self.owned]

    def gather(self):
        """All ranks' owned J / idx assembled in global column-major order (for tests)."""
        t = self.torch
        J, idx = self.owned_J().contiguous(), self.idx.contiguous()
        if self.world == 1:
            return J.cpu().numpy().reshape(-1), idx.cpu().numpy().reshape(-1)
        import torch.distributed as dist
        outJ, outI = [], []
        for r, (b, e) in enumerate(self.parts):
            bj = J if r == self.rank else t.empty((e - b, self.inner), dtype=J.dtype, device=self.device)
            bi = idx if r == self.rank else t.empty((e - b, self.inner), dtype=idx.dtype, device=self.device)
            dist.broadcast(bj, src=r, group=self.group)
            dist.broadcast(bi, src=r, group=self.group)
            outJ.append(bj.cpu().numpy().reshape(-1))
            outI.append(bi.cpu().numpy().reshape(-1))
        return np.concatenate(outJ), np.concatenate(outI)

    def close(self):
        if getattr(self, "_comm_stream", None) is not None:
            self._comm_stream.synchronize()
        if self._rank is not None:
            self._rank.close()
            self._rank = None
