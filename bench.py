#!/usr/bin/env python3
"""Benchmark of the hot path: frames/s (track+map) and rays/s on the Replica room0 work budget,
synthetic 640x480 RGB-D, N GPUs of one node (one process per GPU, RCCL gradient all-reduce).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = one frame-equivalent of the reference's budget (loopy_slam_amd/workload.py):
40 tracking iterations x 1500 rays + 60 mapping iterations x 5000 rays (24 geometry + 36 colour),
each iteration = ray gather, inside-mask, render forward, loss, render backward, Adam - and on every 5th step (a mapped
frame) the frame's point insertion + index rebuild + full-frame render (the headline since round 4; the iterations alone
are reported beside it as ms_per_step_iterations).
Prints ONE JSON line (rank 0).  See DESIGN.md §Measurement for how roofline / cpu_baseline are obtained.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--points', type=int, default=100_000)
    ap.add_argument('--headline-only', action='store_true',
                    help='skip the TUM / ScanNet budgets (`workloads`): profiling passes, whose per-kernel averages must be those of the headline workload')
    ap.add_argument('--strong', action='store_true',
                    help='strong scaling: the per-frame ray budget is SPLIT over the ranks (R / N rays per rank and iteration) instead of '
                         'every rank bringing a full batch.  frames/s of ONE sequence is bounded by the replicated tracking and the track -> map '
                         'dependency (about 2-2.8x at 8 GPUs, DESIGN.md section 6); the 6x-at-8-GPUs target is the weak-scaling rays/s figure')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` on its own: start the N ranks (one process per GPU) the way the driver's torchrun line does and
        # hand its exit code back; rank 0 of that job prints the JSON line
        import socket
        import subprocess
        s = socket.socket()
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
               '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    import torch
    import torch.distributed as dist
    from loopy_slam_amd import core, workload, parallel, profile

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    # testing hooks for a 1-GPU box: all ranks on device 0 over gloo (RCCL refuses two ranks on one device)
    one_dev = os.environ.get('LOOPY_DIST_ONE_DEVICE') == '1'
    backend = os.environ.get('LOOPY_DIST_BACKEND', 'nccl')
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}')
    if rank != 0:
        # rank 0 prints the ONE JSON line; whatever a library of another rank writes to its stdout (this RCCL build prints a version banner
        # when its buffers are flushed at exit) must not land behind it in the launcher's merged output
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
    if world > 1 and not one_dev and torch.cuda.device_count() < world:
        raise SystemExit(f'bench.py: {world} ranks need {world} GPUs, {torch.cuda.device_count()} visible '
                         '(LOOPY_DIST_ONE_DEVICE=1 LOOPY_DIST_BACKEND=gloo puts every rank on device 0 - a functional check, not a measurement)')
    if one_dev:
        local = 0
    torch.cuda.set_device(local)
    # before the process group: the library's side streams take their hardware queues first (lk_streams_init; a side stream that lands on the
    # launch stream's queue behind torch's / RCCL's streams serialises the weight-gradient fork of every 'color' iteration)
    eng = core.Engine()
    dctx = None
    # LOOPY_DIST_FORCE=1 with one rank: the data-parallel code path (lk_map_frame split in phases around a real RCCL all-reduce, bucket
    # pack / unpack, replicated tracking's broadcast) on ONE GPU - what the exchange machinery costs before any second GPU is involved
    force_dist = world == 1 and os.environ.get('LOOPY_DIST_FORCE') == '1'
    if force_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29517')
        dist.init_process_group(backend, rank=0, world_size=1, **({'device_id': torch.device('cuda', local)} if backend == 'nccl' else {}))
        dctx = parallel.DistContext(0, 1)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend)
        dctx = parallel.DistContext(rank, world)
    budget = workload.Budget(n_points=args.points)
    if args.strong and world > 1:
        budget.map_rays = max(32, budget.map_rays // world)          # tracking is replicated (not sharded) in either mode
    wl = workload.FrameWorkload(eng, budget, dist=dctx)
    cloud_dev = tuple(t[:wl.n].clone() for t in (wl.pos, wl.geo, wl.col)) + (wl.n_rooms,) if (world == 1 and not force_dist) else None
    cloud0 = tuple(t[:wl.n].cpu() for t in (wl.pos, wl.geo, wl.col)) if (rank == 0 and not args.no_cpu_baseline) else None

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        wl.step()
    # one extra untimed step with HIP events around EVERY kernel picks the dominant kernel of this workload ...
    barrier()
    prof_all = profile.KernelTimer(eng, '*')
    prof_all.start()
    wl.step()
    barrier()
    kall = prof_all.stop()
    dominant = os.environ.get('BENCH_ROOFLINE_KERNEL') or profile.dominant_kernel(
        {k: v for k, v in kall.items() if k in profile.work_per_step(budget)})
    # ... and the timed region records events on the launch stream around that kernel only.
    # THE TIMED STEP IS THE FULL STEP: 40 tracking + 60 mapping iterations per frame and, on every 5th step (a MAPPED frame of the
    # reference's schedule, mapping.every_frame), what that frame does around its iterations - insertion of 6 000 pixels through the
    # radius test + feature rows + index rebuild (Mapper.py:421-482) and the full-frame render (Mapper.py:966-969)
    wl.frame_no = 0
    wl.step(full=True)                  # untimed: first-use allocations of the full-frame render state
    wl.frame_no = 0
    prof = profile.KernelTimer(eng, dominant)
    barrier()
    prof.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wl.step(full=True)
    barrier()
    dt = time.perf_counter() - t0
    kstat = prof.stop()
    n_mapped = (args.steps + budget.every_frame - 1) // budget.every_frame
    # the same kernel alone on the chip: one more untimed step with the library's second stream switched off
    eng.lib.check(eng.lib.dll.lk_set_serial(1), 'lk_set_serial')
    barrier()
    prof_s = profile.KernelTimer(eng, dominant)
    prof_s.start()
    wl.step()
    barrier()
    kstat_serial = prof_s.stop()
    eng.lib.check(eng.lib.dll.lk_set_serial(0), 'lk_set_serial')
    # beside it: the iterations alone (no mapped-frame extras) - the figure rounds 1-3 reported as `value`
    barrier()
    t0i = time.perf_counter()
    for _ in range(args.steps):
        wl.step()
    barrier()
    dt_iter = time.perf_counter() - t0i
    # PCIe-inclusive: the C ABI takes DEVICE pointers, so a caller that receives its RGB-D frames in host memory uploads one frame
    # (640 x 480 fp32 depth + RGB = 4.9 MB) per step; the same timed steps with that upload from pinned memory in front of each
    # (reported beside `value`, never as `value`)
    host_d, host_c = wl.depth_stack.cpu().pin_memory(), wl.color_stack.cpu().pin_memory()      # every frame of the window: contents unchanged
    wl.frame_no = 0
    barrier()
    t0h = time.perf_counter()
    for _ in range(args.steps):
        k = wl.frame_no % budget.window
        wl.depth_stack[k].copy_(host_d[k], non_blocking=True)
        wl.color_stack[k].copy_(host_c[k], non_blocking=True)
        wl.step(full=True)
    barrier()
    dt_host = time.perf_counter() - t0h
    # The other two single-GPU budgets of BASELINE.json (configs 3-5's 1-GPU content) on the same map: TUM (200 x 5 000 tracking from the
    # gradient-pixel pool + 150 x 10 000 mapping per frame, dynamic radii) and ScanNet (100 x 5 000 + 60 x 10 000, exposure encoding) -
    # 3 timed FULL steps each (their mapped-frame extras included on the budget's own every_frame schedule), reported under `workloads`
    others = {}
    if cloud_dev is not None and not args.headline_only:
        del wl.mapper, wl.tracker
        for name, mk in (('tum', workload.Budget.tum), ('scannet', workload.Budget.scannet)):
            bo = mk(n_points=args.points)
            wo = workload.FrameWorkload(eng, bo, cloud=cloud_dev)
            wo.step(full=True); wo.step()               # untimed: first-use allocations, one plain and one mapped frame
            n_o = 3
            wo.frame_no = 0
            barrier()
            t0o = time.perf_counter()
            for _ in range(n_o):
                wo.step(full=True)
            barrier()
            dto = (time.perf_counter() - t0o) / n_o
            others[name] = {'ms_per_step': 1e3 * dto, 'frames_per_s': 1.0 / dto, 'rays_per_s': bo.rays_per_frame / dto, 'steps': n_o,
                            'rays_per_step': bo.rays_per_frame,
                            'whole_step_fp32_frac': profile.step_flops(bo) / dto / (profile.PEAK_F32_MFMA_TFLOPS * 1e12),
                            'workload': f'{bo.track_iters} track it x {bo.track_rays} rays' + (' (gradient-pixel pool)' if bo.grad_pool else '') +
                                        f' + {bo.map_iters} map it x {bo.map_rays} rays ({bo.map_geo_iters} geometry + {bo.map_iters - bo.map_geo_iters} colour) per frame, '
                                        f'window {bo.window}, ' + ('dynamic radii, ' if bo.dynamic_radius else '') + ('exposure encoding, ' if bo.exposure else '') +
                                        f'mapped-frame extras every {bo.every_frame}th step, N={wo.b.n_points} points'}
            del wo
        # BASELINE configs 4 / 5's single-GPU content at THEIR map sizes: the map grows by rooms at the online density (synthetic.build_cloud_online:
        # one 6 x 4 x 3 m room per 100 000 points, the camera in the last one - the rest is dead weight for the index and the tables, as
        # in a long sequence): (i) the Replica budget's full step on a 2 000 000-point map (config 4's merged cloud), (ii) ONE end-of-sequence
        # refinement call over ALL rows of a 5 000 000-point map with half-precision feature tables (config 5: every row trainable, colour
        # decoder frozen, 10 000 rays per iteration, Mapper.py:884-897; LK_FLAG_FEATS_F16)
        from loopy_slam_amd import steps as _steps, synthetic as _syn

        def grown(n_rooms, seed):
            base = cloud_dev[0][:budget.n_points // 3 * 3]
            pos = torch.cat([base + _syn.room_offset(r, eng.device) for r in range(n_rooms)], 0).contiguous()
            g = torch.Generator(device=eng.device).manual_seed(seed)
            return (pos, 0.1 * torch.randn(pos.shape[0], 32, generator=g, device=eng.device), 0.1 * torch.randn(pos.shape[0], 32, generator=g, device=eng.device), n_rooms)

        big = workload.FrameWorkload(eng, workload.Budget(n_points=budget.n_points * 20), cloud=grown(20, 7))
        big.step(full=True); big.step()
        big.frame_no = 0
        barrier()
        t0o = time.perf_counter()
        for _ in range(5):
            big.step(full=True)
        barrier()
        dto = (time.perf_counter() - t0o) / 5
        others['replica_2m'] = {'ms_per_step': 1e3 * dto, 'frames_per_s': 1.0 / dto, 'rays_per_s': budget.rays_per_frame / dto, 'steps': 5,
                                'rays_per_step': budget.rays_per_frame, 'n_points': big.n,
                                'workload': f'the headline budget (40 x 1500 + 60 x 5000 rays, mapped-frame extras on step 1 of 5) on a {big.n}-point map (20 rooms)'}
        del big
        bo = workload.Budget.tum(n_points=budget.n_points * 50)      # plain colour model, 10 000 mapping rays, per-pixel dynamic radii
        ref = workload.FrameWorkload(eng, bo, cloud=grown(50, 11))
        geo16, col16 = ref.geo.half(), ref.col.half()
        mo = _steps.MapOptimizer(eng, ref.cfg, ref.dec, ref.knn, ref.pos, geo16, col16, None, bo.map_rays, workload.MAP_LRS, w_color=0.1, fix_color_decoder=True)
        n_it, n_geo = 300, 90                 # one optimize_map call of the refinement's size class (mapping.iters 300, geo_iter_ratio 0.3)
        rnd = ref._draws(n_it, bo.map_rays, ref.H * ref.W)
        fid = (torch.arange(bo.map_rays, dtype=torch.int32) % bo.window).to(eng.device)
        log = eng.zeros(n_it, 4)
        for rep in range(2):                  # the second call is timed (the first allocates)
            mo.new_frame(None, None)
            barrier()
            t0o = time.perf_counter()
            mo.run(n_it, n_geo, ref.frames, rnd, fid, (0, ref.H, 0, ref.W), ref.intr, ref.H, ref.W, log)
            barrier()
            dto = time.perf_counter() - t0o
        others['refine_5m_f16'] = {'ms_per_call': 1e3 * dto, 'ms_per_iteration': 1e3 * dto / n_it, 'rays_per_s': n_it * bo.map_rays / dto, 'iterations': n_it,
                                   'n_points': ref.n, 'feature_tables': 'float16 (LK_FLAG_FEATS_F16), gradients / Adam moments fp32',
                                   'workload': f'one whole-map refinement call: {n_geo} geometry + {n_it - n_geo} colour iterations x {bo.map_rays} rays, every row of the '
                                               f'{ref.n}-point map trainable (rows = NULL: only rows that received a gradient are stepped), colour decoder frozen'}
        del mo, ref, geo16, col16
    if world > 1:
        t = torch.tensor([dt, dt_iter, dt_host], device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, dt_iter, dt_host = float(t[0].item()), float(t[1].item()), float(t[2].item())
    rays_per_step = budget.rays_per_frame
    # whole-job rays of a step: every rank brings its own mapping rays; the tracking iterations are REPLICATED on the ranks (the same
    # rays everywhere, no exchange - steps.TrackOptimizer), so they count once however many ranks repeat them
    map_rays_step, track_rays_step = budget.map_iters * budget.map_rays, budget.track_iters * budget.track_rays
    rays_all = map_rays_step * world + track_rays_step
    total_rays = rays_all * args.steps
    out = {
        'metric': 'rays/s (track+map, Replica room0 per-frame budget, 640x480 synthetic RGB-D)' +
                  ('' if world == 1 else (', strong scaling: the frame budget split over the ranks' if args.strong else
                                          ', weak scaling: every rank brings a full frame budget of rays to the shared map')),
        'value': total_rays / dt, 'unit': 'rays/s',
        # every rank works on the SAME frame (the ranks share one map and sum their gradients): a step is one frame whatever N is
        'frames_per_s': args.steps / dt,
        'step': f'the FULL step: every {budget.every_frame}th of the {args.steps} timed steps ({n_mapped} of them) is a mapped frame that also inserts '
                f'{budget.pixels_adding} pixels (lk_add_points + feature rows + lk_knn_build, Mapper.py:421-482) and renders the 640x480 frame '
                f'(307 200 rays, Mapper.py:966-969; these rays are NOT counted in `value`' + (', on a stream of its own beside the next frames\' tracking' if wl.render_stream is not None else '') + f'); {wl.n_added} points added over the run, map {wl.n} points',
        # the iterations alone (what rounds 1-3 reported as the headline)
        'ms_per_step_iterations': 1e3 * dt_iter / args.steps, 'frames_per_s_iterations': args.steps / dt_iter,
        'ms_per_step_host_frames': 1e3 * dt_host / args.steps,
        'host_frames': 'the same steps with one RGB-D frame (4.9 MB, pinned host memory) uploaded over PCIe in front of each - the boundary takes device pointers',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True, 'scaling': 'strong' if (args.strong and world > 1) else 'weak',
        'vs_baseline': None,
        'dtype': 'f32 (decoder products as split products on the 16-bit matrix pipe: fp16x3 forward, bf16x6 / pre-scaled fp16x3 backward, '
                 'fp32 accumulate; weight gradients on pre-scaled fp16x3 too)',
        'data': 'synthetic',
        'config': {'workload': 'Replica room0 budget: 40 track it x 1500 rays + 60 map it x 5000 rays per frame '
                               '(24 geometry + 36 colour) on the frustum rows of the mapped frame, S=5, k=8, C=32, rel-pos colour MLP, '
                               f'N={budget.n_points} points ' + ('laid down by radius-de-duplicated insertion (lk_add_points, r_add 0.04)' if budget.online_cloud else '(random pixels of 24 views)') + ', 640x480 synthetic room',
                   'rays_per_step': rays_per_step, 'rays_per_step_all_ranks': rays_all,
                   'parallelism': f'dp{world} (mapping ray-sharded with one gradient all-reduce per iteration; tracking replicated)'},
    }
    if rank == 0:
        out['roofline'] = profile.roofline(kstat, budget, dominant)
        ser = profile.roofline(kstat_serial, budget, dominant)
        if out['roofline'] is not None and ser is not None:
            out['roofline']['serial'] = {k: ser[k] for k in ('avg_launch_us', 'achieved', 'frac') if k in ser}
            out['roofline']['serial']['note'] = 'same kernel with the side stream off (lk_set_serial): nothing else shares the chip'
        if out['roofline'] is not None:
            out['roofline']['chosen_by'] = ('largest summed duration in the profiled step (kernels within 3 % of it are level and the longer average launch decides); '
                                            'every kernel is in roofline_all_kernels')
        # north_star: fraction of the HBM roofline of the whole step (11.1 KB/ray forward, +20.5 KB/ray backward with the
        # feature-gradient scatter; tracking iterations have no scatter: 2 x 11.1 KB/ray)
        step_bytes = world * (budget.map_iters * budget.map_rays * 31.6e3 + budget.track_iters * budget.track_rays * 22.2e3)
        out['hbm_frac_whole_step'] = step_bytes * args.steps / dt / (world * profile.PEAK_HBM_GBS * 1e9)
        # the same step against the fp32 matrix roof SURVEY 8d names: algorithmic MLP FLOPs (forward + backward-data + weight gradients of
        # every iteration, profile.step_flops) / time / 157.3 TFLOP/s
        if out['roofline'] is not None:
            out['roofline']['whole_step_fp32_frac'] = world * profile.step_flops(budget) * args.steps / dt / (world * profile.PEAK_F32_MFMA_TFLOPS * 1e12)
        if others:
            out['workloads'] = others
        out['host_cores'] = os.cpu_count()
        # every timed kernel against its own roof, from the profiled warm-up step (events around every launch, so slightly slower
        # than the timed region)
        out['roofline_all_kernels'] = []
        for kn in sorted(kall, key=lambda k: -kall[k]['total_ms']):
            rk = profile.roofline(kall, budget, kn)
            if rk is not None:
                out['roofline_all_kernels'].append({k: (round(rk[k], 4) if isinstance(rk[k], float) else rk[k])
                                                    for k in ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'avg_launch_us', 'achieved_from',
                                                              'algorithmic_bytes_per_launch_avg', 'measured_bytes_per_launch_avg', 'frac_of_hbm_peak_measured_bytes',
                                                              'algorithmic_flops_per_launch_avg', 't_mfma_over_t_hbm_measured') if k in rk})
        # the whole 'color' iteration (57 % of the step) against SURVEY 8(d)'s algorithmic bytes, from the committed per-stage counter table
        modes, src_st = profile.stage_traffic()
        if modes and 'color' in modes and out['roofline'] is not None:
            c = modes['color']
            out['roofline']['color_iteration'] = {
                'traffic_mb': round(c['read_mb_per_iteration'] + c['written_mb_per_iteration'], 1), 'algorithmic_mb': c['algorithmic_mb_per_iteration'],
                'traffic_ratio': c['traffic_ratio'], 'source': src_st,
                'note': 'L2-fabric bytes of every launch of one 5 000-ray colour iteration (separate --pmc FETCH_SIZE / WRITE_SIZE passes, tools/stage_traffic.sh) '
                        '/ 5 000 rays x 31.6 KB'}
        out['kernel_ms_per_step'] = {k: round(v['total_ms'], 3) for k, v in sorted(kall.items(), key=lambda kv: -kv[1]['total_ms'])}
        if not args.no_cpu_baseline:
            import bench_cpu_baseline
            out['cpu_baseline'] = bench_cpu_baseline.run(budget, cloud=cloud0)
    if world > 1 or force_dist:
        dist.destroy_process_group()
    if rank == 0:
        # the JSON line is the LAST thing on stdout: C-level buffers first (RCCL's banner sits in libc's buffer until flushed)
        import ctypes
        sys.stdout.flush()
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
