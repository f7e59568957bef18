"""Multi-GPU: one process per GPU, torch.distributed (backend 'nccl' = RCCL over xGMI on ROCm; 'gloo' in CPU tests).

The hot path shards by rays: every rank holds the full map (positions, grid, both feature
tables, decoder blob, Adam state), renders / back-propagates its own ray shard, and the gradients — sums over
rays — are added across ranks with ONE all-reduce per iteration; the identical Adam step then runs on every
rank, so no parameter broadcast is needed.  The all-reduce payload is the decoder-gradient blob plus the two
feature-gradient tables restricted to the rows being optimised (row_index), packed into one bucket."""
import torch
import torch.distributed as dist


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

    def all_reduce_grads(self, mo, stage='color'):
        """mo: steps.MapOptimizer after render_backward.  Sum over ranks exactly what this stage's Adam step consumes:
        the decoder-gradient ranges being stepped (geometry: embedder._B only; colour: + every colour-decoder tensor),
        g_geo[rows], and in the colour stage g_col[rows] - one bucket, one all-reduce."""
        gs = mo.gs
        rows = mo.rows.long() if mo.rows is not None else None
        ranges = _merge(list(mo.geo_dec_ranges) + (list(mo.col_dec_ranges) if stage == 'color' else []))
        tables = [gs.g_geo] + ([gs.g_col] if stage == 'color' else [])
        parts = [gs.g_weights[o:o + n] for o, n in ranges]
        parts += [(t.index_select(0, rows) if rows is not None else t).reshape(-1) for t in tables]
        n = sum(p.numel() for p in parts)
        if self._bucket is None or self._bucket.numel() != n:
            self._bucket = torch.empty(n, dtype=torch.float32, device=parts[0].device)
        torch.cat(parts, out=self._bucket)
        dist.all_reduce(self._bucket, op=dist.ReduceOp.SUM)
        o = 0
        for off, cnt in ranges:
            gs.g_weights[off:off + cnt].copy_(self._bucket[o:o + cnt]); o += cnt
        for t in tables:
            if rows is not None:
                k = rows.numel() * t.shape[1]
                t.index_copy_(0, rows, self._bucket[o:o + k].view(-1, t.shape[1])); o += k
            else:
                k = t.numel()
                t.view(-1).copy_(self._bucket[o:o + k]); o += k

    def all_reduce_vec(self, t):
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t
