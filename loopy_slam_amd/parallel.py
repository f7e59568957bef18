"""Multi-GPU: one process per GPU, torch.distributed (backend 'nccl' = RCCL over xGMI on ROCm; 'gloo' in CPU tests).

The hot path shards by rays: every rank holds the full map (positions, grid, both feature
tables, decoder blob, Adam state), renders / back-propagates its own ray shard, and the gradients — sums over
rays — are added across ranks with ONE all-reduce per iteration; the identical Adam step then runs on every
rank, so no parameter broadcast is needed.  The all-reduce payload is the decoder-gradient blob plus the two
feature-gradient tables restricted to the rows being optimised (row_index), packed into one bucket."""
import torch
import torch.distributed as dist


class DistContext:
    def __init__(self, rank, world):
        self.rank, self.world = rank, world
        self._bucket = None

    def all_reduce_grads(self, mo):
        """mo: steps.MapOptimizer after render_backward.  Sum g_weights, g_geo[rows], g_col[rows] over ranks."""
        gs = mo.gs
        rows = mo.rows.long() if mo.rows is not None else None
        parts = [gs.g_weights]
        if rows is not None:
            parts += [gs.g_geo.index_select(0, rows).reshape(-1), gs.g_col.index_select(0, rows).reshape(-1)]
        else:
            parts += [gs.g_geo.reshape(-1), gs.g_col.reshape(-1)]
        n = sum(p.numel() for p in parts)
        if self._bucket is None or self._bucket.numel() != n:
            self._bucket = torch.empty(n, dtype=torch.float32, device=parts[0].device)
        torch.cat(parts, out=self._bucket)
        dist.all_reduce(self._bucket, op=dist.ReduceOp.SUM)
        o = 0
        gs.g_weights.copy_(self._bucket[o:o + gs.g_weights.numel()]); o += gs.g_weights.numel()
        if rows is not None:
            k = rows.numel() * 32
            gs.g_geo.index_copy_(0, rows, self._bucket[o:o + k].view(-1, 32)); o += k
            gs.g_col.index_copy_(0, rows, self._bucket[o:o + k].view(-1, 32)); o += k
        else:
            k = gs.g_geo.numel()
            gs.g_geo.view(-1).copy_(self._bucket[o:o + k]); o += k
            gs.g_col.view(-1).copy_(self._bucket[o:o + k]); o += k

    def all_reduce_vec(self, t):
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t
