"""Magnitude range of the backward's gradient operands (d h, d hid rows, d c) in a mapper colour iteration of the bench."""
import sys
sys.path.insert(0, '.')
import torch
from loopy_slam_amd import core, workload
eng = core.Engine()
wl = workload.FrameWorkload(eng, workload.Budget())
wl.step()
mo = wl.mapper
P = mo.R * 5
s = mo.gs.scratch
o_dc = (4 + 32) * P
o_dh = (4 + 32 + 32 + 4 + 4 + 4 + 4 + 8 + 8 + 4) * P + ((P + 31) // 32 + 3) // 4 * 288 + ((P + 3) // 4 + 3) // 4 * 32 + 128 * P + 256 * P + P
o_rows = o_dh + 640 * P
for name, t in (('d_raw', s[0:4 * P]), ('dc_col', s[o_dc:o_dc + 32 * P]), ('dh_col', s[o_dh:o_dh + 640 * P]), ('rows dhid', s[o_rows:o_rows + 1536 * P].reshape(-1, 192)[:, :128])):
    a = t.abs().reshape(-1).float()
    nz = a[a > 0]
    q = torch.quantile(nz[torch.randint(0, nz.numel(), (1_000_000,), device=nz.device)], torch.tensor([0.01, 0.5, 0.99], device=nz.device))
    print(f'{name:10s} max {float(a.max()):.3e}  q01 {float(q[0]):.3e} median {float(q[1]):.3e} q99 {float(q[2]):.3e}  zeros {float((a == 0).float().mean()):.3f}')
