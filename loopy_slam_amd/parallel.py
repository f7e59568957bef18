"""Multi-GPU: one process per GPU, torch.distributed (backend 'nccl' = RCCL over xGMI on ROCm; 'gloo' in CPU tests).

The hot path shards by rays: every rank holds the full map (positions, grid, both feature
tables, decoder blob, Adam state), renders / back-propagates its own ray shard, and the gradients — sums over
rays — are added across ranks with ONE all-reduce per iteration; the identical Adam step then runs on every
rank, so no parameter broadcast is needed.  The all-reduce payload is the decoder-gradient blob plus the two
feature-gradient tables restricted to the rows being optimised (row_index), packed into one bucket."""
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


class DistContext:
    def __init__(self, rank, world):
        self.rank, self.world = rank, world
        self._bucket = None
        self._touched = None
        self._keep = None

    def all_reduce_grads(self, mo, stage='color'):
        """mo: steps.MapOptimizer after render_backward.  Sum over ranks exactly what this stage's Adam step consumes:
        the decoder-gradient ranges being stepped (geometry: embedder._B only; colour: + every colour-decoder tensor),
        g_geo[rows], and in the colour stage g_col[rows] - one bucket, one all-reduce, one pack and one unpack launch
        (lk_bucket_copy: the torch formulation was eight small kernels per iteration)."""
        gs, eng = mo.gs, mo.eng
        ranges = _merge(list(mo.geo_dec_ranges) + (list(mo.col_dec_ranges) if stage == 'color' else []))
        tables = [gs.g_geo] + ([gs.g_col] if stage == 'color' else [])
        # exposure encoding: d loss / d affine [F,12] is a sum over rays too and rides in the same bucket (the exposure MLP's
        # backward and Adam step run after the exchange, on identical inputs on every rank)
        xs = mo.exposure if stage == 'color' else None
        segs = (_ffi.CopySeg * (len(ranges) + len(tables) + (1 if xs is not None else 0)))()
        n = 0
        for k, (o, cnt) in enumerate(ranges):
            segs[k].data, segs[k].n, segs[k].row_index, segs[k].row_len = ptr(gs.g_weights[o:o + cnt]), cnt, None, 1
            n += cnt
        rows = mo.rows if mo.rows is not None else self.touched_rows(mo)
        for k, t in enumerate(tables, start=len(ranges)):
            segs[k].data, segs[k].n = ptr(t), rows.numel() * t.shape[1]
            segs[k].row_index, segs[k].row_len = ptr(rows), t.shape[1]
            n += segs[k].n
        self._keep = rows
        if xs is not None:
            k = len(ranges) + len(tables)
            segs[k].data, segs[k].n, segs[k].row_index, segs[k].row_len = ptr(xs.g_aff), xs.g_aff.numel(), None, 1
            n += segs[k].n
        if self._bucket is None or self._bucket.numel() != n:
            self._bucket = torch.empty(n, dtype=torch.float32, device=gs.g_weights.device)
        eng.lib.check(eng.lib.dll.lk_bucket_copy(segs, len(segs), ptr(self._bucket), 0, eng.stream), 'lk_bucket_copy')
        dist.all_reduce(self._bucket, op=dist.ReduceOp.SUM)
        eng.lib.check(eng.lib.dll.lk_bucket_copy(segs, len(segs), ptr(self._bucket), 1, eng.stream), 'lk_bucket_copy')

    def touched_rows(self, mo):
        """Whole-map optimisation (rows = None: the final refinement, Mapper.py:884-897): a batch of R rays touches at most
        8 R S rows of the N-row tables, the rest of both gradient tables is exactly zero on every rank.  Exchanging the tables
        themselves would be 256 B x N per iteration (1.28 GB at 5 M points); instead the ranks agree on the UNION of
        the rows they touched - one MAX all-reduce of an N-byte flag vector (5 MB) - and only those rows ride in the bucket.
        Returns the sorted int32 row list (identical on every rank)."""
        N = mo.geo.shape[0]
        if self._touched is None or self._touched.numel() != N:
            self._touched = torch.zeros(N, dtype=torch.uint8, device=mo.geo.device)
        else:
            self._touched.zero_()
        idx = mo.current_nbr_idx().reshape(-1)
        self._touched[idx[idx >= 0].long()] = 1
        dist.all_reduce(self._touched, op=dist.ReduceOp.MAX)
        return torch.nonzero(self._touched).reshape(-1).to(torch.int32)

    def all_reduce_vec(self, t):
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t
