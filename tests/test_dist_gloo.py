"""Multi-GPU path, world_size 2 over gloo: on the CPU emulator, and (-m gpu) with BOTH RANKS ON ONE MI355X through the real
library (RCCL refuses two ranks on one device, so the collectives are staged through the host there; everything else - the
kernels, lk_map_frame split in phases around the exchange, the side-stream row agreement - is the production path).
Ray-sharded data parallel — every rank renders/back-propagates its shard, ONE all-reduce sums {decoder-blob, selected
feature-row} gradients, the same Adam step runs everywhere (loopy_slam_amd/parallel.py).  Checked against a single process that
sees both shards in one batch."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

HH, WW = 24, 32
INTR = (40.0, 40.0, 15.5, 11.5)
R, ITERS = 64, 3
LRS = {'geometry': (0.001, 0.03, 0.0), 'color': (0.005, 0.005, 0.005)}
STAGES = ['geometry', 'color', 'color']


def build(eng, R_batch, dctx, all_rows=False, exposure=False):
    """exposure: the ScanNet model (model.encode_exposure, plain colour MLP) - the 'color' iterations render logits and the loss applies the
    frame's affine; mlp_exposure and the frame's exposure feature are parameters too."""
    from loopy_slam_amd import core, steps, synthetic as syn
    from test_steps_parity import mini_scene, _exposure_module
    c2w, depth_img, color_img, pos, geo, col = mini_scene(3)
    depth_img = depth_img.clone()
    depth_img.reshape(-1)[5] = 1.0                       # no outlier: the inside mask keeps every positive depth
    W = syn.default_weights(seed=9, rel_pos=not exposure, exposure=exposure)
    dec = core.DecoderBlob(eng).pack(W)
    pos_d, geo_d, col_d = eng.f32(pos), eng.f32(geo).clone(), eng.f32(col).clone()
    knn = core.KnnIndex(eng, capacity=pos.shape[0])
    knn.build(pos_d)
    rows = torch.arange(0, pos.shape[0], 3, dtype=torch.int32)
    xp = None
    if exposure:
        xp = (_exposure_module(W).to(eng.device), [(0.2 * torch.ones(8)).to(eng.device).requires_grad_(True)])
    mo = steps.MapOptimizer(eng, core.RenderCfg(rel_pos=not exposure, exposure=exposure), dec, knn, pos_d, geo_d, col_d,
                            None if all_rows else rows.to(eng.device), R_batch, LRS, w_color=0.1, dist=dctx, exposure=xp)
    mo.begin_frame()
    frames = (eng.f32(depth_img).reshape(1, HH, WW), eng.f32(color_img).reshape(1, HH, WW, 3), eng.f32(c2w).reshape(1, 4, 4), None)
    return mo, frames, dec, geo_d, col_d


def draws():
    g = torch.Generator().manual_seed(21)
    return torch.randint(0, HH * WW, (ITERS, 2, R), generator=g, dtype=torch.int32)


def worker(rank, port, q, all_rows=False, native=True, backend='emu', overlap=False, exposure=False):
    try:
        if overlap:          # the row part of the bucket on a communication stream beside the backward's tail (parallel._exchange)
            os.environ['LOOPY_DIST_OVERLAP'] = '1'
        _worker(rank, port, q, all_rows, native, backend, exposure)
    except Exception:                                   # surface the reason in the parent instead of a bare exit code
        import traceback
        q.put(('error', rank, traceback.format_exc()))
        raise


def _worker(rank, port, q, all_rows, native, backend, exposure=False):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=2)
    from loopy_slam_amd import parallel
    from util import make_engine
    eng = make_engine(backend)
    mo, frames, dec, geo_d, col_d = build(eng, R, parallel.DistContext(rank, 2), all_rows, exposure)
    rnd = draws().to(eng.device)
    fid = torch.zeros(R, dtype=torch.int32, device=eng.device)
    losses = []
    if native:          # the native loop (lk_map_frame) split around the exchange; all_rows: touched-row bucket, agreed one iteration ahead
        log = eng.zeros(ITERS, 4)
        mo.run(ITERS, 1, frames, rnd[:, rank].contiguous(), fid, (0, HH, 0, WW), INTR, HH, WW, log)
        log = log.cpu()
        dist.all_reduce(log)
        losses = [float(x) for x in log[:, 0]]
        if exposure:            # every loss row of the call, every column (loss, depth term, colour term, rays): summed over the ranks
            losses = log.reshape(-1).tolist()
    else:
        for it in range(ITERS):
            out4 = mo.iterate(STAGES[it], frames, rnd[it, rank].contiguous(), fid, (0, HH, 0, WW), INTR, HH, WW)
            t = out4.cpu().clone()
            dist.all_reduce(t)
            losses.append(float(t[0]))
    # numpy arrays travel through the queue by value (a tensor travels as a handle the receiver must fetch while the sender lives)
    if rank == 0:
        q.put((losses, dec.blob.cpu().numpy().copy(), geo_d.cpu().numpy().copy(), col_d.cpu().numpy().copy()))
    else:
        q.put((losses, dec.blob.cpu().numpy().copy(), None, None))
    dist.barrier()
    dist.destroy_process_group()


from util import backends  # noqa: E402


@pytest.mark.parametrize('backend', backends())
@pytest.mark.parametrize('all_rows,native', ((False, False), (False, True), (True, True), (False, 'overlap')))
def test_two_rank_grad_allreduce_matches_single_process(all_rows, native, backend):
    """native: the two-rank side runs lk_map_frame split in phases 1 / 2 around the all-reduce (else the per-statement path).
    all_rows: every row of the map is a parameter (final refinement) - the ranks exchange the union of the touched rows, agreed
    one iteration ahead on a side stream.  backend hip: both ranks on cuda:0."""
    from util import make_engine
    overlap = native == 'overlap'        # the native loop with the overlapped exchange (the default of a multi-rank RCCL run)
    native = bool(native)
    torch.set_num_threads(1)
    eng = make_engine(backend)
    mo, frames, dec, geo_d, col_d = build(eng, 2 * R, None, all_rows)
    rnd = draws().to(eng.device)
    fid = torch.zeros(2 * R, dtype=torch.int32, device=eng.device)
    ref_losses = []
    for it in range(ITERS):
        out4 = mo.iterate(STAGES[it], frames, rnd[it].reshape(-1).contiguous(), fid, (0, HH, 0, WW), INTR, HH, WW)
        ref_losses.append(float(out4[0]))
    res = _spawn_two_ranks((all_rows, native, backend, overlap))
    res = [tuple(torch.from_numpy(x) if isinstance(x, np.ndarray) else x for x in r) for r in res]
    full = [r for r in res if r[2] is not None][0]
    other = [r for r in res if r[2] is None][0]
    np.testing.assert_allclose(full[0], ref_losses, rtol=1e-5)
    assert torch.equal(full[1], other[1])                                   # identical parameters on both ranks
    np.testing.assert_allclose(full[1].numpy(), dec.blob.cpu().numpy(), rtol=0, atol=2e-5)
    geo_d, col_d = geo_d.cpu(), col_d.cpu()
    err = (full[2] - geo_d).abs()
    assert float(torch.quantile(err.reshape(-1), 0.999)) < 2e-5 and float(err.max()) < 0.015
    err = (full[3] - col_d).abs()
    assert float(torch.quantile(err.reshape(-1), 0.999)) < 2e-5 and float(err.max()) < 0.0025


def _spawn_two_ranks(args):
    """Two worker processes over gloo (one retry: the free-port probe / gloo rendezvous can lose a race) -> their queue results."""
    for attempt in range(2):
        s = socket.socket()
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
        s.close()
        ctx = mp.get_context('spawn')
        q = ctx.Queue()
        procs = [ctx.Process(target=worker, args=(r, port, q) + tuple(args)) for r in range(2)]
        for p in procs:
            p.start()
        try:
            res = [q.get(timeout=600) for _ in range(2)]
        except Exception as e:
            res = [('error', -1, repr(e))]
        for p in procs:
            p.join(timeout=120)
            if p.is_alive():
                p.kill()
        errs = [r for r in res if r[0] == 'error']
        if not errs and all(p.exitcode == 0 for p in procs):
            return res
        if attempt == 1:
            raise AssertionError(f'workers failed: {errs} exit codes {[p.exitcode for p in procs]}')


@pytest.mark.parametrize('backend', backends())
def test_two_rank_exposure_loop_writes_every_loss_row(backend):
    """Exposure encoding through the phase-split native loop (one lk_map_frame call per phase around the exchange): the 'geometry'
    iterations leave their loss row as per-tile terms that ONE launch sums - behind the sequence's LAST iteration, which with exposure
    encoding is a 'color' iteration that leaves no terms itself (round-4 advisor: the rows [0, n_geo) were never written then).  Every
    row of the log, summed over the two ranks, against the single-process native loop that sees both shards; parameters alike."""
    from util import make_engine
    torch.set_num_threads(1)
    eng = make_engine(backend)
    mo, frames, dec, geo_d, col_d = build(eng, 2 * R, None, False, exposure=True)
    rnd = draws().to(eng.device)
    fid = torch.zeros(2 * R, dtype=torch.int32, device=eng.device)
    log = eng.zeros(ITERS, 4)
    mo.run(ITERS, 1, frames, rnd.reshape(ITERS, -1).contiguous(), fid, (0, HH, 0, WW), INTR, HH, WW, log)
    mo.finish()
    ref = log.cpu().reshape(-1).numpy()
    assert ref.reshape(ITERS, 4)[0, 0] > 0 and ref.reshape(ITERS, 4)[0, 2] == 0 and ref.reshape(ITERS, 4)[1, 2] > 0      # geometry row, then colour rows
    res = _spawn_two_ranks((False, True, backend, False, True))
    for r in res:
        np.testing.assert_allclose(np.array(r[0]), ref, rtol=2e-5, atol=1e-6)
    a, b = (torch.from_numpy(r[1]) for r in res)
    assert torch.equal(a, b)                                               # identical decoders on both ranks
    np.testing.assert_allclose(a.numpy(), dec.blob.cpu().numpy(), rtol=0, atol=2e-5)


def _rccl_worker(port, q, all_rows, mode='direct'):
    try:
        os.environ['MASTER_ADDR'] = '127.0.0.1'
        os.environ['MASTER_PORT'] = str(port)
        # mode: 'direct' = the library's own RCCL communicator on the launch stream (loopy_slam_amd/rccl.py, the default), 'torch' = torch's
        # process-group collectives, 'overlap' = direct + the row part of the bucket on a communication stream beside the backward's tail
        if mode == 'torch':
            os.environ['LOOPY_DIST_TORCH'] = '1'
        if mode in ('overlap', 'overlap-serial'):
            os.environ['LOOPY_DIST_OVERLAP'] = '1'
        torch.cuda.set_device(0)
        from loopy_slam_amd import parallel
        from util import make_engine
        eng = make_engine('hip')                 # before the process group: the library's streams take their hardware queues first
        if mode == 'overlap-serial':
            # the library on the launch stream only (LK_SERIAL / a failed side-stream creation): the communication stream of the overlapped
            # exchange still gets a real dependency on the backward - the rows event is recorded whatever the stream mode (round-4 advisor)
            eng.lib.check(eng.lib.dll.lk_set_serial(1), 'lk_set_serial')
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
        dctx = parallel.DistContext(0, 1)
        mo, frames, dec, geo_d, col_d = build(eng, 2 * R, dctx, all_rows)
        rnd = draws().to(eng.device)
        fid = torch.zeros(2 * R, dtype=torch.int32, device=eng.device)
        log = eng.zeros(ITERS, 4)
        mo.run(ITERS, 1, frames, rnd.reshape(ITERS, -1).contiguous(), fid, (0, HH, 0, WW), INTR, HH, WW, log)
        # the tracker's pose broadcast and a vector sum go over the same communicator
        v = torch.arange(7, dtype=torch.float32, device=eng.device)
        dctx.broadcast(v, src=0)
        dctx.all_reduce_vec(v)
        torch.cuda.synchronize()
        used = 'direct' if dctx._rccl else 'torch'
        q.put(([float(x) for x in log[:, 0].cpu()], dec.blob.cpu().numpy().copy(), geo_d.cpu().numpy().copy(), col_d.cpu().numpy().copy(),
               v.cpu().numpy().copy(), dist.get_backend(), used))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put(('error', traceback.format_exc()))
        raise


@pytest.mark.gpu
@pytest.mark.parametrize('all_rows,mode', ((False, 'direct'), (True, 'direct'), (False, 'torch'), (False, 'overlap'), (False, 'overlap-serial')))
def test_rccl_collectives_on_the_launch_stream(all_rows, mode):
    """The production exchange on the production backend: ONE rank, backend 'nccl' (= RCCL), the native loop split in phases around
    dist.all_reduce of the gradient bucket on DEVICE memory (no host staging), the uint8 MAX agreement on the touched rows and the
    pose broadcast.  A sum over one rank is the identity, so parameters and losses must equal the plain single-process loop - what
    this pins on hardware is that RCCL loads, takes the library's buffers and orders itself against the launch stream on both sides
    of the collective.  (Two or more ranks need as many GPUs: RCCL refuses two ranks on one device - the 2-rank logic runs over gloo
    above, the 8-rank curve is the driver's.)"""
    from util import make_engine
    eng = make_engine('hip')
    mo, frames, dec, geo_d, col_d = build(eng, 2 * R, None, all_rows)
    rnd = draws().to(eng.device)
    fid = torch.zeros(2 * R, dtype=torch.int32, device=eng.device)
    log = eng.zeros(ITERS, 4)
    mo.run(ITERS, 1, frames, rnd.reshape(ITERS, -1).contiguous(), fid, (0, HH, 0, WW), INTR, HH, WW, log)
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(port, q, all_rows, mode))
    p.start()
    res = q.get(timeout=600)
    p.join(timeout=120)
    if p.is_alive():
        p.kill()
    assert res[0] != 'error', res[1]
    assert p.exitcode == 0 and res[5] == 'nccl' and res[6] == ('torch' if mode == 'torch' else 'direct')
    np.testing.assert_allclose(res[0], log[:, 0].cpu().numpy(), rtol=1e-6)
    np.testing.assert_allclose(res[1], dec.blob.cpu().numpy(), rtol=0, atol=2e-6)
    # (feature rows: float atomics order inside the gather is the only source of difference between two runs of the same loop)
    assert float(np.abs(res[2] - geo_d.cpu().numpy()).max()) < 1e-4 and float(np.abs(res[3] - col_d.cpu().numpy()).max()) < 1e-4
    np.testing.assert_array_equal(res[4], np.arange(7, dtype=np.float32))
