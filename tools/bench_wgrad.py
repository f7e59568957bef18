#!/usr/bin/env python3
"""Micro-benchmark of lk_wgrad_single on the GPU: one job at a time, various shapes/chunks."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from loopy_slam_amd import core
from loopy_slam_amd._ffi import ptr

eng = core.Engine()
rows = int(os.environ.get('ROWS', 25000))
for (N, K, mode) in ((128, 128, 0), (128, 128, 1), (128, 168, 1), (128, 32, 0), (128, 40, 1), (32, 128, 2), (128, 52, 0)):
    r = rows * 8 if mode == 2 or K == 52 else rows
    lda = 640 if mode != 2 else 32
    A = torch.randn(r, lda, device='cuda')
    A2 = torch.rand(r, 640 if mode != 2 else 1, device='cuda') * 0.05
    B = torch.randn(r, 640 if K != 52 else 320, device='cuda')
    dW = torch.zeros(N, K + 4, device='cuda'); db = torch.zeros(N, device='cuda')
    for chunk in [int(c) for c in os.environ.get('CHUNKS', '128,256,512,1024').split(',')]:
        for _ in range(2):
            eng.lib.dll.lk_wgrad_single(ptr(A), lda, mode, ptr(A2), A2.shape[1], ptr(B), B.shape[1], N, K, r, ptr(dW), K + 4, ptr(db), chunk, eng.stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            eng.lib.dll.lk_wgrad_single(ptr(A), lda, mode, ptr(A2), A2.shape[1], ptr(B), B.shape[1], N, K, r, ptr(dW), K + 4, ptr(db), chunk, eng.stream)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        gb = r * (N * (2 if mode == 1 else 1) + K) * 4 / 1e9
        print(f'N={N:3d} K={K:3d} mode={mode} rows={r:6d} chunk={chunk:4d}: {us:7.1f} us  {2*N*K*r/us/1e6:6.2f} TFLOP/s  {gb/us*1e6:6.0f} GB/s')
