"""How far is the fp32 CPU oracle from its own float64 evaluation on the ray gradients of the BA-mode graph (tracker-mode render + mapper loss,\n5 000 rays, 100 000 points)?  python tools/probe/oracle_noise_ba.py {replica|tum}  ->  ~1e-2 max-norm: the bar of tests/test_parity_at_size.py::\ntest_ba_mode_backward_at_bench_size cannot be tighter than the summation-order noise of the fp32 graph itself."""
import sys, numpy as np, torch, time
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import atsize as A
from oracle import hotpath as H
from loopy_slam_amd import synthetic as syn
torch.set_num_threads(8)
rel = sys.argv[1] == 'replica'
R = 5000
pos, geo, col = A.scene(100_000)
W = syn.default_weights(rel_pos=rel)
b = A.ray_batch(R, frame=7, holes=0.0, seed=3)
z, _ = H.sample_z(b['gt_depth'], 0.98, 1.02, 0.3, 5)
p = H.sample_points(b['rays_o'], b['rays_d'], z)
kn = A.contract_knn(pos, p, np.float32(0.08 ** 2))[:3]
def run(dt):
    Wr = {k: v.to(dt) for k, v in W.items()}
    ro, rd = b["rays_o"].to(dt).clone().detach().requires_grad_(True), b["rays_d"].to(dt).clone().detach().requires_grad_(True)
    o = H.render_batch(A.ocfg(rel), ro, rd, b['gt_depth'].to(dt), pos.to(dt), geo.to(dt), col.to(dt), Wr, 'color', tracker=True, knn=kn)
    loss = H.mapper_loss(o['depth'], o['color'], o['valid_ray'], b['gt_depth'].to(dt), b['gt_color'].to(dt), 'color', 0.1)
    loss[0].backward()
    return ro.grad, rd.grad
t=time.time()
g32 = run(torch.float32)
try:
    g64 = run(torch.float64)
    for n, a, c in (('rays_o', g32[0], g64[0]), ('rays_d', g32[1], g64[1])):
        print(n, 'oracle fp32 vs fp64: max-norm rel %.3g' % float((a.double() - c).abs().max() / c.abs().max()))
except Exception as e:
    print('fp64 oracle failed:', repr(e)[:300])
print(time.time()-t)
