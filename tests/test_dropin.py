"""The drop-in seam (/root/reference/run.py:10-49, src/Point_SLAM.py:94,177-209): run.py with the reference's command
line, Point_SLAM.load_pretrain with the reference's key handling, and the `src.*` import aliases."""
import copy
import os
import subprocess
import sys
import warnings

import pytest
import torch

from loopy_slam_amd import config, slam, synthetic as syn
from util import make_engine, backends
from test_slam_api import mini_cfg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pretrained_file(path, seed=99):
    """A stand-in for pretrained/middle_fine.pt: ckpt['model'] of a ConvONet run - `decoder.coarse.*` (the middle decoder = our geometry
    decoder's names), `decoder.fine.*`, `encoder.*`, and a coarse key the geometry decoder does not have."""
    W = syn.default_weights(seed)
    g = torch.Generator().manual_seed(seed)
    W = {k: (v + 0.01 * torch.randn(v.shape, generator=g) if k.endswith('bias') else v) for k, v in W.items()}     # (random init has zero biases)
    model = {}
    for k, v in W.items():
        if k.startswith('geo_decoder.'):
            model['decoder.coarse.' + k[len('geo_decoder.'):]] = v.clone()
            model['decoder.fine.' + k[len('geo_decoder.'):]] = v.clone() * 3.0
    model['decoder.coarse.not_in_this_decoder.weight'] = torch.ones(4, 4)
    model['encoder.coarse.unet.weight'] = torch.ones(8, 8)
    model['encoder_decoder_bridge.weight'] = torch.ones(2)           # 'decoder' AND 'encoder' in the key: skipped by the reference's test
    torch.save({'model': model}, path)
    return W


@pytest.mark.parametrize('backend', backends())
def test_load_pretrain_takes_the_middle_decoder(backend, tmp_path):
    eng = make_engine(backend)
    cfg = mini_cfg()
    path = str(tmp_path / 'middle_fine.pt')
    Wp = _pretrained_file(path)
    cfg['pretrained_decoders'] = {'middle_fine': path}
    with warnings.catch_warnings():
        warnings.simplefilter('error')                              # a found file must not warn
        ps = slam.Point_SLAM(cfg, None, eng=eng)
    sd = ps.shared_decoders.state_dict()
    W0 = syn.default_weights(cfg['setup_seed'], rel_pos=cfg['model']['encode_rel_pos_in_col'])
    n_geo = 0
    for k, v in sd.items():
        if k.startswith('geo_decoder.') and k in Wp:
            assert torch.equal(v.reshape(Wp[k].shape), Wp[k]), k                   # the coarse decoder's values, not the fine ones
            assert not torch.equal(v.reshape(W0[k].shape), W0[k]), k
            n_geo += 1
        elif k.startswith('color_decoder.') and k in W0:
            assert torch.equal(v.reshape(W0[k].shape), W0[k]), k                   # the colour decoder keeps its initialisation
    assert n_geo == 23 and len(ps.pretrained_loaded) >= 23            # 5 x (linear + fc_c) x (weight, bias) + output + B
    assert not any('not_in_this_decoder' in k or 'encoder' in k for k in sd)
    # the kernels read the loaded decoder: the fragments were repacked (a render differs from one with the random decoder)
    idx, color, depth, c2w = ps.frame_reader[0]
    ps.mapper.map_frame(0, color, depth, c2w, cur_c2w=c2w)
    assert torch.isfinite(ps.mapper.last_log).all()
    # size mismatch raises, strict or not (torch semantics)
    bad = {'pts_linears.0.weight': torch.zeros(3, 3)}
    with pytest.raises(RuntimeError):
        ps.shared_decoders.geo_decoder.load_state_dict(bad, strict=False)


def test_missing_pretrained_file_warns_and_keeps_random_init():
    eng = make_engine('emu')
    cfg = mini_cfg()
    cfg['pretrained_decoders'] = {'middle_fine': '/nonexistent/middle_fine.pt'}
    with pytest.warns(UserWarning, match='middle_fine'):
        ps = slam.Point_SLAM(cfg, None, eng=eng)
    assert ps.pretrained_loaded == []


def test_src_aliases_import_the_native_classes():
    code = ('from src import config; from src.Point_SLAM import Point_SLAM; from src.common import setup_seed; '
            'from src.neural_point import NeuralPointCloud; from src.utils.Renderer import Renderer; '
            'import loopy_slam_amd.slam as S; assert Point_SLAM is S.Point_SLAM and NeuralPointCloud is S.NeuralPointCloud; '
            'setup_seed(1219); print(config.load_config("configs/Replica/room0.yaml", "configs/point_slam.yaml")["mapping"]["iters"])')
    out = subprocess.run([sys.executable, '-c', code], cwd=ROOT, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip() == '300'


def test_stop_ends_the_run_after_frame_n():
    eng = make_engine('emu')
    cfg = mini_cfg()
    cfg['data']['n_frames'] = 9
    cfg['stop'] = 2
    ps = slam.Point_SLAM(cfg, None, eng=eng)
    est, gt = ps.run()
    assert est.shape[0] == 3 and int(ps.idx[0]) == 2


@pytest.mark.gpu
def test_run_py_with_the_reference_command_line(tmp_path):
    """`python run.py configs/Synthetic/room.yaml --stop 10` on the GPU box: eleven frames through the native loops, a checkpoint in the
    reference's layout at frame 10 that Logger.load restores."""
    out = str(tmp_path / 'out')
    r = subprocess.run([sys.executable, 'run.py', 'configs/Synthetic/room.yaml', '--stop', '10', '--output', out], cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert 'run.py: 11 frames' in r.stdout, r.stdout[-2000:]
    ck = os.path.join(out, 'ckpts', '00010.tar')
    assert os.path.exists(ck)
    d = torch.load(ck, map_location='cpu', weights_only=False)
    assert d['idx'] == 10 and len(d['keyframe_list']) >= 2 and d['pts_num'] > 1000
    for k in ('cloud_pos', 'decoder_state_dict', 'gt_c2w_list', 'estimate_c2w_list', 'keyframe_dict', 'selected_keyframes'):
        assert k in d
