"""ctypes binding of libloopyhip.so (C ABI: include/loopy_hip.h).

The product path loads exactly one library: the in-tree gfx950 build
``loopy_slam_amd/libloopyhip.so`` (built by ``loopy_slam_amd/csrc/build.py`` /
``__graft_entry__.build()``).  There is no CPU fallback: if the library is missing
or no HIP device is visible, ``get_lib()`` raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libloopyhip.so')

LK_K = 8
LK_C = 32
LK_S_MAX = 8

FLAG_STAGE_COLOR = 1 << 0
FLAG_TRACKER = 1 << 1
FLAG_REL_POS = 1 << 2
FLAG_COLOR_LOGITS = 1 << 3
FLAG_SAVE_ACT = 1 << 4
FLAG_GRAD_FEATS = 1 << 5
FLAG_GRAD_WEIGHTS = 1 << 6
FLAG_GRAD_RAYS = 1 << 7
FLAG_ALL_DEPTH_POS = 1 << 8
FLAG_ZERO_ABSENT = 1 << 9
FLAG_MAPPER_LOSS = 1 << 10
FLAG_Z_GIVEN = 1 << 12
FLAG_FEATS_F16 = 1 << 13
FLAG_UNIT_LOSS_GRADS = 1 << 11
FLAG_EMBED_GRADS_ONLY = 1 << 14
FLAG_GRAD_GEO_DECODER = 1 << 15
FLAG_CHECK_RANGE = 1 << 16
STATUS_WEIGHT_RANGE, STATUS_ACT_RANGE = 1, 2
ERR_RANGE = -4

EXPOSURE_MAX_F = 32
ADAM_MAX_SEG = 16

_fp = C.c_void_p     # every device pointer crosses the ABI as an integer address


class WeightEntry(C.Structure):
    _fields_ = [('name', C.c_char * 64), ('offset', C.c_int64), ('rows', C.c_int32), ('cols', C.c_int32),
                ('ld', C.c_int32), ('col_split', C.c_int32), ('col_shift', C.c_int32)]


class RenderDesc(C.Structure):
    _fields_ = [
        ('R', C.c_int32), ('S', C.c_int32), ('stats_chunk', C.c_int32), ('flags', C.c_uint32),
        ('rays_o', _fp), ('rays_d', _fp), ('gt_depth', _fp), ('r2_ray', _fp),
        ('knn', _fp), ('pos', _fp), ('geo_feats', _fp), ('col_feats', _fp), ('weights', _fp), ('weights_frag', _fp),
        ('affine', _fp), ('noise_geo', _fp), ('noise_col', _fp),
        ('near_surface', C.c_float), ('far_surface', C.c_float), ('near_end', C.c_float), ('coef', C.c_float),
        ('r2_static', C.c_float), ('min_nn', C.c_int32),
        ('depth', _fp), ('var', _fp), ('color', _fp), ('valid_ray', _fp),
        ('z', _fp), ('nbr_idx', _fp), ('nbr_w', _fp), ('nbr_count', _fp), ('c_geo', _fp), ('c_col', _fp),
        ('raw', _fp), ('far_stats', _fp), ('act', _fp),
        ('d_depth', _fp), ('d_var', _fp), ('d_color', _fp),
        ('g_geo_feats', _fp), ('g_col_feats', _fp), ('g_weights', _fp), ('g_rays_o', _fp), ('g_rays_d', _fp),
        ('g_affine', _fp), ('grad_row_mask', _fp), ('bwd_scratch', _fp), ('bwd_scratch_cap', C.c_int64),
        ('loss_gt_color', _fp), ('loss_out4', _fp), ('loss_w_color', C.c_float),
    ]


class AdamSeg(C.Structure):
    _fields_ = [('p', _fp), ('g', _fp), ('m', _fp), ('v', _fp), ('n', C.c_int64), ('lr', C.c_float),
                ('step', C.c_int32), ('row_index', _fp), ('row_len', C.c_int32), ('zero_grad', C.c_int32), ('p_f16', C.c_int32),
                ('row_flags', _fp), ('g_compact', C.c_int32)]


class CopySeg(C.Structure):
    _fields_ = [('data', _fp), ('n', C.c_int64), ('row_index', _fp), ('row_len', C.c_int32)]


MAX_SPANS = 8


class BlobSpan(C.Structure):
    _fields_ = [('offset', C.c_int64), ('n', C.c_int64)]


EXPOSURE_GRAD_FLOATS = 1024 + 128 + 1536 + 12 + EXPOSURE_MAX_F * 8


class ExposureDesc(C.Structure):
    """lk_exposure_desc (include/loopy_hip.h)."""
    _fields_ = [('feats', _fp), ('W1', _fp), ('b1', _fp), ('W2', _fp), ('b2', _fp), ('F', C.c_int32),
                ('aff', _fp), ('hid', _fp), ('g_aff', _fp), ('g', _fp), ('adam', _fp),
                ('lr_mlp', C.c_float), ('lr_feat', C.c_float), ('feat_first', C.c_int32), ('feat_count', C.c_int32), ('bwd_scale', _fp)]


class TrackDesc(C.Structure):
    """lk_track_desc (include/loopy_hip.h)."""
    _fields_ = [
        ('render', RenderDesc),
        ('depth_img', _fp), ('color_img', _fp), ('r2_map', _fp),
        ('H', C.c_int32), ('W', C.c_int32), ('H0', C.c_int32), ('W0', C.c_int32), ('w', C.c_int32),
        ('fx', C.c_float), ('fy', C.c_float), ('cx', C.c_float), ('cy', C.c_float),
        ('rnd', _fp), ('gt_color', _fp), ('pix_i', _fp), ('pix_j', _fp), ('thr', _fp), ('scratch_u32', _fp), ('loss_scratch', _fp),
        ('cam7', _fp), ('g_cam7', _fp), ('adam_mv', _fp),
        ('lr_T', C.c_float), ('lr_q', C.c_float), ('w_color', C.c_float), ('use_color', C.c_int32), ('hist_post', C.c_int32),
        ('hist', _fp), ('log', _fp), ('iters', C.c_int32), ('work', _fp), ('exposure', C.POINTER(ExposureDesc)),
    ]


class MapDesc(C.Structure):
    """lk_map_desc (include/loopy_hip.h)."""
    _fields_ = [
        ('render', RenderDesc),
        ('depth_stack', _fp), ('color_stack', _fp), ('c2w_stack', _fp), ('c2w_stride', C.c_int32), ('r2_map_stack', _fp),
        ('frame_id', _fp), ('rnd', _fp),
        ('H', C.c_int32), ('W', C.c_int32), ('H0', C.c_int32), ('W0', C.c_int32), ('w', C.c_int32),
        ('fx', C.c_float), ('fy', C.c_float), ('cx', C.c_float), ('cy', C.c_float),
        ('gt_color', _fp), ('thr', _fp), ('scratch_u32', _fp),
        ('w_color', C.c_float), ('log', _fp),
        ('weights_rw', _fp), ('weights_frag_rw', _fp), ('geo_feats_rw', _fp), ('col_feats_rw', _fp),
        ('rows', _fp), ('n_rows', C.c_int64), ('adam_rows', _fp),
        ('geo_dec', BlobSpan * MAX_SPANS), ('n_geo_dec', C.c_int32),
        ('col_dec', BlobSpan * MAX_SPANS), ('n_col_dec', C.c_int32),
        ('adam_dec', _fp),
        ('lr', (C.c_float * 3) * 2),
        ('iters', C.c_int32), ('n_geo_iters', C.c_int32), ('work', _fp), ('exposure', C.POINTER(ExposureDesc)),
        ('grad_bucket', _fp), ('bucket_geo_dec', C.c_int64 * MAX_SPANS), ('bucket_col_dec', C.c_int64 * MAX_SPANS),
        ('bucket_geo_rows', C.c_int64), ('bucket_col_rows', C.c_int64),
        ('union_rows_flagged', C.c_int32), ('it_offset', C.c_int32), ('batches_ready', C.c_int32), ('train_geo_decoder', C.c_int32), ('signal_rows', C.c_int32),
    ]


class LoopyError(RuntimeError):
    pass


class LoopyLib:
    """Typed handle on one loaded libloopyhip build."""

    def __init__(self, path):
        if not os.path.exists(path):
            raise LoopyError(f'{path} not found: build it with `python loopy_slam_amd/csrc/build.py` '
                             f'(or __graft_entry__.build()); there is no CPU fallback')
        self.path = path
        # torch must initialise ITS HIP runtime first: loading libloopyhip.so before torch would pull
        # in /opt/rocm's libamdhip64 as a second runtime instance in the process (no device visible to it).
        import torch  # noqa: F401
        self.dll = C.CDLL(path)
        d = self.dll
        d.lk_version.restype = C.c_int
        d.lk_last_error.restype = C.c_char_p
        d.lk_knn_create.argtypes = [C.c_float, C.c_int64, C.c_int64, C.POINTER(C.c_void_p)]
        d.lk_knn_destroy.argtypes = [C.c_void_p]
        d.lk_knn_build.argtypes = [C.c_void_p, _fp, C.c_int64, C.c_void_p]
        d.lk_knn_append.argtypes = [C.c_void_p, _fp, C.c_int64, C.c_void_p]
        d.lk_knn_size.argtypes = [C.c_void_p]
        d.lk_knn_size.restype = C.c_int64
        d.lk_knn_query.argtypes = [C.c_void_p, _fp, C.c_int64, C.c_float, _fp, _fp, _fp, _fp, C.c_void_p]
        d.lk_weight_layout.argtypes = [C.POINTER(WeightEntry), C.c_int]
        d.lk_weight_blob_floats.restype = C.c_int64
        d.lk_weight_frag_floats.restype = C.c_int64
        d.lk_weights_repack.argtypes = [_fp, _fp, C.c_void_p]
        d.lk_render_act_floats.argtypes = [C.c_int32, C.c_int32, C.c_uint32]
        d.lk_render_act_floats.restype = C.c_int64
        d.lk_render_fwd.argtypes = [C.POINTER(RenderDesc), C.c_void_p]
        for name, argtypes, restype in (
            ('lk_render_bwd_scratch_floats', [C.c_int32, C.c_int32, C.c_uint32], C.c_int64),
            ('lk_render_bwd', [C.POINTER(RenderDesc), C.c_void_p], C.c_int),
            ('lk_loss_mapper', [C.c_int32, _fp, _fp, _fp, _fp, _fp, C.c_float, C.c_int32, _fp, _fp, _fp, C.c_void_p], C.c_int),
            ('lk_loss_tracker', [C.c_int32, _fp, _fp, _fp, _fp, _fp, C.c_float, C.c_int32, _fp, _fp, _fp, _fp, C.c_void_p], C.c_int),
            ('lk_adam_step', [C.POINTER(AdamSeg), C.c_int32, C.c_float, C.c_float, C.c_float, C.c_void_p], C.c_int),
            ('lk_bucket_copy', [C.POINTER(CopySeg), C.c_int32, C.c_void_p, C.c_int32, C.c_void_p], C.c_int),
            ('lk_exposure_fwd', [_fp] * 5 + [C.c_int32, _fp, _fp, C.c_void_p], C.c_int),
            ('lk_exposure_bwd', [_fp] * 5 + [C.c_int32, _fp, C.c_void_p], C.c_int),
            ('lk_loss_mapper_exposure', [C.c_int32] + [_fp] * 7 + [C.c_int32, C.c_float, _fp, _fp, _fp, _fp, C.c_void_p], C.c_int),
            ('lk_rays_from_pose', [_fp, _fp, _fp, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, _fp, _fp, C.c_void_p], C.c_int),
            ('lk_pose_bwd', [_fp, _fp, _fp, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, _fp, _fp, _fp, C.c_void_p], C.c_int),
            ('lk_gather_rays', [_fp, _fp, _fp, C.c_int32, _fp, _fp, _fp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, _fp, _fp, _fp, _fp, _fp, _fp, _fp,
                                C.c_void_p], C.c_int),
            ('lk_frustum_rows', [_fp, C.c_int32, C.POINTER(C.c_float), _fp, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float,
                                 C.c_float, C.c_int32, _fp, _fp, _fp, _fp, _fp, C.c_void_p], C.c_int),
            ('lk_add_points', [C.c_void_p, _fp, _fp, _fp, C.c_int32, C.c_float, _fp, C.c_float, C.c_float, C.c_int32,
                               _fp, _fp, _fp, _fp, C.c_void_p], C.c_int),
            ('lk_radius_maps', [_fp, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double, _fp, _fp, _fp, C.c_void_p], C.c_int),
            ('lk_top_grad_pixels', [_fp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _fp, C.c_int32,
                                    _fp, _fp, C.c_void_p], C.c_int),
            ('lk_wgrad_single', [_fp, C.c_int32, C.c_int32, _fp, C.c_int32, _fp, C.c_int32, C.c_int32, C.c_int32, C.c_int64,
                                 _fp, C.c_int32, _fp, C.c_int32, C.c_void_p], C.c_int),
            ('lk_profile_begin', [C.c_char_p], C.c_int),
            ('lk_profile_end', [C.c_char_p, C.c_int], C.c_int),
            ('lk_compact', [_fp, C.c_int32, _fp, _fp, C.c_void_p], C.c_int),
            ('lk_set_serial', [C.c_int32], C.c_int),
            ('lk_streams_init', [], C.c_int),
            ('lk_debug_occupancy', [C.POINTER(C.c_int32)], C.c_int),
            ('lk_debug_side_delay', [C.c_int32], C.c_int),
            ('lk_weights_repack_checked', [_fp, _fp, C.c_void_p], C.c_int),
            ('lk_status_peek', [C.POINTER(C.c_uint32)], C.c_int),
            ('lk_status_sync', [C.c_void_p, C.POINTER(C.c_uint32)], C.c_int),
            ('lk_status_clear', [], C.c_int),
            ('lk_track_work_floats', [C.c_int32, C.c_int32, C.c_int32], C.c_int64),
            ('lk_map_work_floats', [C.c_int32, C.c_int32, C.c_int32], C.c_int64),
            ('lk_map_work_nbr_idx', [C.c_int32, C.c_int32, C.c_int32], C.c_int64),
            ('lk_track_frame', [C.POINTER(TrackDesc), C.c_void_p], C.c_int),
            ('lk_map_frame', [C.POINTER(MapDesc), C.c_int32, C.c_int32, C.c_int32, C.c_void_p], C.c_int),
            ('lk_map_prepare', [C.POINTER(MapDesc), C.c_void_p], C.c_int),
            ('lk_inside_mask', [_fp, C.c_int32, _fp, _fp, _fp, _fp, C.c_void_p], C.c_int),
            ('lk_compact_large', [_fp, C.c_int32, _fp, _fp, _fp, C.c_void_p], C.c_int),
            ('lk_touch_rows', [_fp, C.c_int64, _fp, C.c_int32, C.c_void_p], C.c_int),
            ('lk_knn_flag_rows', [C.c_void_p, _fp, C.c_int64, C.c_void_p], C.c_int),
            ('lk_map_wait_lists', [C.POINTER(MapDesc), C.c_int32, C.c_void_p], C.c_int),
            ('lk_map_wait_rows', [C.POINTER(MapDesc), C.c_void_p], C.c_int),
        ):
            if hasattr(d, name):
                fn = getattr(d, name)
                fn.argtypes = argtypes
                fn.restype = restype
        if d.lk_version() != 1:
            raise LoopyError(f'{path}: ABI version {d.lk_version()} != 1')

    def check(self, rc, what=''):
        if rc != 0:
            raise LoopyError(f'{what} failed (rc={rc}): {self.dll.lk_last_error().decode()}')

    def weight_layout(self):
        n = self.dll.lk_weight_layout(None, 0)
        arr = (WeightEntry * n)()
        self.dll.lk_weight_layout(arr, n)
        return [dict(name=e.name.decode(), offset=e.offset, rows=e.rows, cols=e.cols, ld=e.ld,
                     col_split=e.col_split, col_shift=e.col_shift) for e in arr]

    def blob_floats(self):
        return int(self.dll.lk_weight_blob_floats())


_LIB = None


def get_lib():
    """The product library.  Raises if it is not built or no GPU is visible (no CPU path)."""
    global _LIB
    if _LIB is None:
        import torch
        if not torch.cuda.is_available():
            raise LoopyError('loopy_slam_amd needs a HIP device (MI355X); none is visible and there is no CPU fallback')
        _LIB = LoopyLib(LIB_PATH)
    return _LIB


def ptr(t):
    """Device address of a tensor (or 0 for None)."""
    return 0 if t is None else t.data_ptr()
