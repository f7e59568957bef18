"""Multi-GPU: one process per GPU, torch.distributed (backend 'nccl' = RCCL over xGMI on ROCm; 'gloo' in CPU tests).

The hot path shards by rays: every rank holds the full map (positions, grid, both feature
tables, decoder blob, Adam state), renders / back-propagates its own ray shard, and the gradients — sums over
rays — are added across ranks with ONE all-reduce per iteration; the identical Adam step then runs on every
rank, so no parameter broadcast is needed.  The all-reduce payload is the decoder-gradient blob plus the two
feature-gradient tables restricted to the rows being optimised (row_index), packed into one bucket.

No host synchronisation of the launch stream inside an iteration: when the whole map is optimised the row list of an
iteration (the union of the rows the ranks' batches touch) is agreed ONE ITERATION AHEAD on a side stream
(lk_map_frame's neighbour search runs ahead of its loop, so the lists of iteration it + 1 exist while iteration it
renders), and the host only ever waits for that side stream's event."""
import ctypes as C
import os

import torch
import torch.distributed as dist

from . import _ffi
from ._ffi import ptr


def _merge(ranges, gap=4096):
    """(offset, n) ranges -> few contiguous spans (alignment padding between neighbouring tensors rides along: one copy
    kernel per span instead of one per tensor)."""
    out = []
    for off, n in sorted(ranges):
        if out and off - (out[-1][0] + out[-1][1]) <= gap:
            out[-1] = (out[-1][0], off + n - out[-1][0])
        else:
            out.append((off, n))
    return out


def ray_range(n_rays, rank, world, chunk):
    """The contiguous ray range [lo, hi) of `rank` when n_rays rays are shared out over `world` ranks in whole groups of `chunk` rays
    ("contiguous ray ranges (tracking, render_img)").  Group-aligned because the reference evaluates far_bb per batch of
    ray_batch_size rays (/root/reference/src/utils/Renderer.py:102-121, 241-266): a rank that starts on a group boundary forms the same
    groups as one process does, so its rays come out bit for bit as in the full-frame render."""
    groups = (n_rays + chunk - 1) // chunk
    g0, g1 = (groups * rank) // world, (groups * (rank + 1)) // world
    return min(n_rays, g0 * chunk), min(n_rays, g1 * chunk)


class _RowAgreement:
    """One in-flight agreement on the touched rows of an iteration: device list + count on its way to pinned host memory."""

    def __init__(self, eng, N):
        dev = eng.device
        self.flags = torch.zeros(N, dtype=torch.uint8, device=dev)
        self.rows = torch.empty(max(N, 1), dtype=torch.int32, device=dev)
        self.count = torch.zeros(1, dtype=torch.int32, device=dev)
        self.scratch = torch.empty((N + 255) // 256 + 1, dtype=torch.int32, device=dev)
        self.count_host = torch.zeros(1, dtype=torch.int32)
        if dev.type == 'cuda':
            self.count_host = self.count_host.pin_memory()
        self.event = torch.cuda.Event() if dev.type == 'cuda' else None
        self.it = -1


class DistContext:
    def __init__(self, rank, world):
        self.rank, self.world = rank, world
        self._bucket = None
        self._keep = None
        self._seg_cache = None
        self._agree = {}            # slot (it & 1) -> _RowAgreement
        self._side = None
        self._main_ev = None
        # the row part of the gradient bucket on a communication stream beside the backward's tail (_exchange): with ONE rank it costs two event
        # hand-overs and hides nothing (+1.2 ms per step, profiles/r4_ab_dist_exchange.txt) - with peers on the wire it hides the larger of the two collectives (3-4 MB of
        # feature-row gradients: tens of microseconds of ring time per iteration against ~35 us of hand-overs).  Default: on when ranks exchange over
        # RCCL, off otherwise; LOOPY_DIST_OVERLAP=0 / 1 overrides
        ov = os.environ.get('LOOPY_DIST_OVERLAP')
        self.overlap = (ov == '1') if ov is not None else (world > 1 and dist.is_initialized() and dist.get_backend() == 'nccl')
        self._comm = None
        self._comm_ev = None
        self._rccl = None

    # ------------------------------------------------------------------ collectives
    def _direct(self, t):
        """The library's own RCCL communicator (loopy_slam_amd/rccl.py: collectives enqueued on the launch stream itself - no second stream,
        no event hand-overs) for device tensors when the process group runs over RCCL; LOOPY_DIST_TORCH=1 keeps torch's collectives (A/B).
        Created at the first collective - every rank reaches it at the same point of the program."""
        if not t.is_cuda or dist.get_backend() != 'nccl' or os.environ.get('LOOPY_DIST_TORCH') == '1':
            return None
        if self._rccl is None:
            from . import rccl
            # the ranks decide TOGETHER: a rank whose librccl does not load must not walk on to torch's collectives while the others sit in the
            # communicator's id broadcast / ncclCommInitRank.  Step 1: can every rank load the library (MIN over the torch group)?
            try:
                rccl._load()
                err = None
            except OSError as e:
                err = e
            if not self._all_ok(err is None, t.device):
                self._fallback(err or 'librccl missing on another rank')
                return None
            # step 2: the communicator itself (collective; a rank that fails here raises out of a call its peers are blocked in - RCCL's own
            # failure mode, not recoverable by agreement), then one more agreement so that all ranks use it or none does
            try:
                comm, err = rccl.RcclComm(self.rank, self.world, t.device), None
            except Exception as e:          # noqa: BLE001
                comm, err = None, e
            if not self._all_ok(comm is not None, t.device):
                if comm is not None:
                    comm.close()
                self._fallback(err or 'communicator creation failed on another rank')
                return None
            self._rccl = comm
        return self._rccl or None

    def _all_ok(self, ok, device):
        """True iff `ok` on EVERY rank (MIN all-reduce over the torch process group; a single rank decides alone)."""
        if self.world <= 1:
            return bool(ok)
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(int(flag.item()))

    def _fallback(self, why):
        import warnings
        warnings.warn(f'direct RCCL communicator unavailable ({why}): collectives go through torch.distributed on every rank')
        self._rccl = False

    def _all_reduce(self, t, op):
        """All-reduce in place; on a backend without device collectives (gloo with HIP tensors: the 2-ranks-on-one-GPU test
        hook - RCCL refuses two ranks on one device) the tensor is staged through the host."""
        # the direct communicator carries what the hot path exchanges - contiguous float32 SUM and uint8 MAX; anything else is torch's
        kind = 'sum' if op == dist.ReduceOp.SUM else ('max' if op == dist.ReduceOp.MAX else None)
        r = self._direct(t) if (kind is not None and t.is_contiguous() and ((kind == 'sum' and t.dtype == torch.float32) or
                                                                             (kind == 'max' and t.dtype == torch.uint8))) else None
        if r is not None:
            r.all_reduce(t, kind)
            return
        if t.is_cuda and dist.get_backend() != 'nccl':
            h = t.cpu()
            dist.all_reduce(h, op=op)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=op)

    def broadcast(self, t, src=0):
        r = self._direct(t)
        if r is not None and t.dtype in (torch.float32, torch.uint8) and t.is_contiguous():
            return r.broadcast(t, src)
        if t.is_cuda and dist.get_backend() != 'nccl':
            h = t.cpu()
            dist.broadcast(h, src=src)
            t.copy_(h)
        else:
            dist.broadcast(t, src=src)
        return t

    def all_reduce_vec(self, t):
        self._all_reduce(t, dist.ReduceOp.SUM)
        return t

    def gather_ranges(self, tensors, lo, hi):
        """Full-frame outputs from per-rank ray ranges: every tensor holds this rank's rays [lo, hi) along dim 0 and ZEROS elsewhere; one SUM
        all-reduce over the concatenation leaves the whole frame on every rank (x + 0 = x exactly; ranges are disjoint).  One collective of
        5 floats per ray (6 MB for 640 x 480) instead of an all-gather per tensor with ragged shards - the float32 SUM is what both the direct
        RCCL communicator and the gloo test path carry."""
        if self.world <= 1:
            return tensors
        flat = torch.cat([t.reshape(t.shape[0], -1).float() for t in tensors], dim=1).contiguous()
        self._all_reduce(flat, dist.ReduceOp.SUM)
        out, c = [], 0
        for t in tensors:
            w = t[0].numel() if t.shape[0] else 1
            out.append(flat[:, c:c + w].reshape(t.shape).to(t.dtype))
            c += w
        return out

    # ------------------------------------------------------------------ gradient bucket
    def all_reduce_grads(self, mo, stage='color', it=None, desc=None):
        """mo: steps.MapOptimizer after render_backward.  Sum over ranks exactly what this stage's Adam step consumes:
        the decoder-gradient ranges being stepped (geometry: embedder._B only; colour: + every colour-decoder tensor),
        g_geo[rows], and in the colour stage g_col[rows] - one bucket, one all-reduce, one pack and one unpack launch
        (lk_bucket_copy: the torch formulation was eight small kernels per iteration).
        desc: the lk_map_desc of a phase-split lk_map_frame loop over a row list without exposure encoding - the step of the phase-2
        call then reads the summed gradients from the bucket itself (lk_map_desc::grad_bucket): the pack clears its sources
        (lk_bucket_copy mode 2) and nothing is unpacked."""
        gs, eng = mo.gs, mo.eng
        direct = desc is not None and mo.rows is not None and (mo.exposure is None)
        if desc is not None and not direct:
            desc.grad_bucket = None
        rows = mo.rows if mo.rows is not None else self.touched_rows(mo, it)
        # the segment table of a stage is the same for every iteration of an optimize_map call (same buffers, same row list): built
        # once - at 72-us 'geometry' iterations the interpreter time of rebuilding it per iteration was the longer side
        # keyed by the optimiser's call token (steps.MapOptimizer.call_token: a process-wide counter bumped by every constructor and
        # new_frame) - device addresses and id()s are reused by the allocator / CPython across optimize_map calls, a key made of them
        # could hit with another call's exposure buffers or decoder ranges; the entry keeps what its segment table points at alive
        key = (stage, mo.call_token, rows.data_ptr(), rows.numel())
        if self._seg_cache is not None and self._seg_cache[0] == key:
            _, segs, n, offs = self._seg_cache
            bucket = self._bucket[:n]
            self._exchange(mo, desc, stage, segs, bucket, offs, direct)
            return
        ranges = _merge(list(mo.geo_dec_ranges) + (list(mo.col_dec_ranges) if stage == 'color' else []))
        tables = [gs.g_geo] + ([gs.g_col] if stage == 'color' else [])
        # exposure encoding: d loss / d affine [F,12] is a sum over rays too and rides in the same bucket (the exposure MLP's
        # backward and Adam step run after the exchange, on identical inputs on every rank)
        xs = mo.exposure if stage == 'color' else None
        segs = (_ffi.CopySeg * (len(ranges) + len(tables) + (1 if xs is not None else 0)))()
        n = 0
        offs = dict(ranges=[], tables=[])          # where every segment starts in the bucket (floats)
        for k, (o, cnt) in enumerate(ranges):
            segs[k].data, segs[k].n, segs[k].row_index, segs[k].row_len = ptr(gs.g_weights[o:o + cnt]), cnt, None, 1
            offs['ranges'].append((o, cnt, n))
            n += cnt
        for k, t in enumerate(tables, start=len(ranges)):
            segs[k].data, segs[k].n = ptr(t), rows.numel() * t.shape[1]
            segs[k].row_index, segs[k].row_len = ptr(rows), t.shape[1]
            offs['tables'].append(n)
            n += segs[k].n
        self._keep = (rows, gs.g_weights, gs.g_geo, gs.g_col, xs.g_aff if xs is not None else None, mo)
        if xs is not None:
            k = len(ranges) + len(tables)
            segs[k].data, segs[k].n, segs[k].row_index, segs[k].row_len = ptr(xs.g_aff), xs.g_aff.numel(), None, 1
            n += segs[k].n
        if self._bucket is None or self._bucket.numel() < n:
            self._bucket = torch.empty(max(n, 2 * (self._bucket.numel() if self._bucket is not None else 0)),
                                       dtype=torch.float32, device=gs.g_weights.device)
        bucket = self._bucket[:n]
        offs['n_ranges'], offs['n_dec'] = len(ranges), (offs['tables'][0] if offs['tables'] else n)
        self._seg_cache = (key, segs, n, offs) if mo.rows is not None else None       # touched-row lists change every iteration
        self._exchange(mo, desc, stage, segs, bucket, offs, direct)

    def overlaps_rows(self, mo):
        """True when the iterations of `mo` exchange their feature-row gradients beside the tail of the backward (lk_map_desc::signal_rows)."""
        return self.overlap and mo.rows is not None and mo.exposure is None and mo.eng.device.type == 'cuda' and self.world >= 1

    def _exchange(self, mo, desc, stage, segs, bucket, offs, direct):
        """Pack, all-reduce, (unpack).  The bucket is [decoder spans | rows of the geometry table | rows of the colour table].  The row part
        is final as soon as the feature-row gather of the backward has run (k_feat_gather), while the weight-gradient launch and the
        reduction of the decoder gradients (k_wgrad, k_bwd_reduce: 50-130 us of a 'color' iteration) are still to come: with a phase-split
        lk_map_frame loop over a row list (`direct`) the row part is packed and all-reduced on a COMMUNICATION stream that waits for the
        library's rows event (lk_map_wait_rows) only - beside that tail - and the launch stream packs and all-reduces the (small) decoder
        part behind the reduction, then waits for the communication stream.  RCCL runs the two collectives in the order they were issued
        (rows first) on every rank."""
        eng = mo.eng
        dll = eng.lib.dll
        if not (direct and self.overlaps_rows(mo)):
            eng.lib.check(dll.lk_bucket_copy(segs, len(segs), ptr(bucket), 2 if direct else 0, eng.stream), 'lk_bucket_copy')
            self._all_reduce(bucket, dist.ReduceOp.SUM)
            if direct:
                self._point_desc_at_bucket(mo, desc, stage, bucket, offs)
            else:
                eng.lib.check(dll.lk_bucket_copy(segs, len(segs), ptr(bucket), 1, eng.stream), 'lk_bucket_copy')
            return
        nr, nd = offs['n_ranges'], offs['n_dec']
        if self._comm is None:
            self._comm = torch.cuda.Stream(eng.device)
            self._comm_ev = torch.cuda.Event()
        seg_t = type(segs[0])
        row_segs = (seg_t * (len(segs) - nr)).from_address(C.addressof(segs) + nr * C.sizeof(seg_t))
        with torch.cuda.stream(self._comm):
            eng.lib.check(dll.lk_map_wait_rows(C.byref(desc), eng.stream), 'lk_map_wait_rows')
            eng.lib.check(dll.lk_bucket_copy(row_segs, len(segs) - nr, ptr(bucket[nd:]), 2, eng.stream), 'lk_bucket_copy')
            self._all_reduce(bucket[nd:], dist.ReduceOp.SUM)
            self._comm_ev.record(self._comm)
        eng.lib.check(dll.lk_bucket_copy(segs, nr, ptr(bucket[:nd]), 2, eng.stream), 'lk_bucket_copy')
        self._all_reduce(bucket[:nd], dist.ReduceOp.SUM)
        torch.cuda.current_stream(eng.device).wait_event(self._comm_ev)
        self._point_desc_at_bucket(mo, desc, stage, bucket, offs)

    @staticmethod
    def _point_desc_at_bucket(mo, desc, stage, bucket, offs):
        """lk_map_desc::grad_bucket and the bucket offsets of every span / table of this stage's step."""
        def bucket_off(span_off):
            for o, cnt, b in offs['ranges']:
                if o <= span_off < o + cnt:
                    return b + (span_off - o)
            raise AssertionError('decoder span outside the bucket')
        desc.grad_bucket = ptr(bucket)
        for k, (o, _) in enumerate(mo.geo_dec_ranges):
            desc.bucket_geo_dec[k] = bucket_off(o)
        if stage == 'color':
            for k, (o, _) in enumerate(mo.col_dec_ranges):
                desc.bucket_col_dec[k] = bucket_off(o)
        desc.bucket_geo_rows = offs['tables'][0]
        desc.bucket_col_rows = offs['tables'][1] if len(offs['tables']) > 1 else 0

    # ------------------------------------------------------------------ touched rows (whole-map optimisation)
    def _enqueue_agreement(self, mo, it, idx):
        """Enqueue, on the current stream: flags of the rows iteration `it`'s lists touch -> MAX over the ranks -> compaction
        -> the count on its way to the host.  Whole-map optimisation (rows = None: the final refinement, Mapper.py:884-897):
        a batch of R rays touches at most 8 R S rows of the N-row tables, the rest of both gradient tables is exactly zero
        on every rank.  Exchanging the tables themselves would be 256 B x N per iteration (1.28 GB at 5 M points, SURVEY
        §8e); instead the ranks agree on the UNION of the rows they touched - one MAX all-reduce of an N-byte flag vector
        (5 MB) - and only those rows ride in the bucket."""
        eng = mo.eng
        N = mo.geo.shape[0]
        ag = self._agree.get(it & 1)
        if ag is None or ag.flags.numel() != N:
            ag = self._agree[it & 1] = _RowAgreement(eng, N)
        ag.it = it
        ag.flags.zero_()
        dll = eng.lib.dll
        eng.lib.check(dll.lk_touch_rows(ptr(idx), idx.numel(), ptr(ag.flags), N, eng.stream), 'lk_touch_rows')
        self._all_reduce(ag.flags, dist.ReduceOp.MAX)
        eng.lib.check(dll.lk_compact_large(ptr(ag.flags), N, ptr(ag.rows), ptr(ag.count), ptr(ag.scratch), eng.stream), 'lk_compact_large')
        ag.count_host.copy_(ag.count, non_blocking=True)
        if ag.event is not None:
            ag.event.record(torch.cuda.current_stream(eng.device))
        return ag

    def prefetch_touched(self, mo, it):
        """Called right after the phase-1 call of iteration it - 1 has been enqueued: agree on iteration `it`'s rows on the side
        stream, beside that iteration's render."""
        eng = mo.eng
        if eng.device.type != 'cuda':
            self._enqueue_agreement(mo, it, mo.nbr_idx_of(it))
            return
        if self._side is None:
            self._side = torch.cuda.Stream(eng.device)
            self._main_ev = torch.cuda.Event()
        main = torch.cuda.current_stream(eng.device)
        self._main_ev.record(main)                  # serial mode of the library: the lists are written in the main stream's order
        with torch.cuda.stream(self._side):
            self._side.wait_event(self._main_ev)
            mo.wait_lists(it)                       # look-ahead mode: they are written on the library's third stream
            self._enqueue_agreement(mo, it, mo.nbr_idx_of(it))

    def touched_rows(self, mo, it=None):
        """Sorted int32 row list (identical on every rank) of iteration `it`: the prefetched agreement if there is one - the
        host waits for the SIDE stream's event only, the launch stream keeps its queue - else (first iteration of a call, or
        the per-statement path whose lists exist only after its own forward) computed in line with one read-back."""
        ag = self._agree.get(it & 1) if it is not None else None
        if ag is not None and ag.it == it:
            if ag.event is not None:
                ag.event.synchronize()
                torch.cuda.current_stream(mo.eng.device).wait_event(ag.event)
        else:
            ag = self._enqueue_agreement(mo, it if it is not None else 0, mo.nbr_idx_of(it))
            if ag.event is not None:
                ag.event.synchronize()
        ag.it = -1
        self._last_ag = ag
        return ag.rows[:int(ag.count_host.item())]

    def flag_union(self, mo):
        """After the exchange of a whole-map iteration: the rows ANY rank touched join the index's row flags (lk_knn_flag_rows), so that
        lk_map_frame's step visits them - a row only another rank touched has just received its gradient through the all-reduce."""
        ag = getattr(self, '_last_ag', None)
        if ag is None:
            return
        eng = mo.eng
        eng.lib.check(eng.lib.dll.lk_knn_flag_rows(mo.knn.h, ptr(ag.flags), ag.flags.numel(), eng.stream), 'lk_knn_flag_rows')
