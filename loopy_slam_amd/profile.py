"""Per-kernel timing (lk_profile_*: HIP events on the launch stream) and the roofline bookkeeping
bench.py reports.  Peaks from /opt/skills/guides/MI355X_MICROARCH.md: fp32 matrix (v_mfma_f32_32x32x2_f32)
157.3 TFLOP/s, HBM3E 8 TB/s."""
import ctypes as C

PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0

# algorithmic multiply-adds per SAMPLE POINT of k_decode_bwd (backward-data of both decoders):
#   colour: W4^T,W2^T,W1^T (3 x 128x128) + W3^T hidden part (128x128) + 5 x U^T (128x32) + Wo^T (3x128)
#   geometry: W4^T,W2^T,W1^T (3 x 32x32) + W3^T (32x128) + W0^T (32x96) + 5 x U^T (32x32) + wo (32)
MAC_DECODE_BWD_COLOR = 3 * 128 * 128 + 128 * 128 + 5 * 128 * 32 + 3 * 128
MAC_DECODE_BWD_GEO = 3 * 32 * 32 + 32 * 128 + 32 * 96 + 5 * 32 * 32 + 32
# tracker mode adds the embedding columns of the colour skip / first layer: W3^T (128x40) + W0^T (128x40)
MAC_DECODE_BWD_TRACK_EXTRA = 2 * 128 * 40


class KernelTimer:
    def __init__(self, eng, names='*'):
        self.eng, self.names = eng, names

    def start(self):
        self.eng.lib.check(self.eng.lib.dll.lk_profile_begin(self.names.encode()), 'lk_profile_begin')

    def stop(self):
        buf = C.create_string_buffer(4096)
        self.eng.lib.check(self.eng.lib.dll.lk_profile_end(buf, 4096), 'lk_profile_end')
        out = {}
        for line in buf.value.decode().splitlines():
            name, calls, ms = line.split()
            out[name] = dict(calls=int(calls), total_ms=float(ms))
        return out


def roofline_decode_bwd(kstat, budget, steps=1):
    """achieved = algorithmic FLOPs of all k_decode_bwd launches of the timed region / their summed duration."""
    k = kstat.get('k_decode_bwd')
    if not k or k['total_ms'] <= 0:
        return None
    S = 5
    macs_per_step = (budget.map_geo_iters * budget.map_rays * S * MAC_DECODE_BWD_GEO +
                     (budget.map_iters - budget.map_geo_iters) * budget.map_rays * S * (MAC_DECODE_BWD_GEO + MAC_DECODE_BWD_COLOR) +
                     budget.track_iters * budget.track_rays * S * (MAC_DECODE_BWD_GEO + MAC_DECODE_BWD_COLOR + MAC_DECODE_BWD_TRACK_EXTRA))
    launches_per_step = budget.map_iters + budget.track_iters
    n_steps = k['calls'] / launches_per_step
    flops = 2.0 * macs_per_step * n_steps
    achieved = flops / (k['total_ms'] * 1e-3) / 1e12
    return {'kernel': 'k_decode_bwd', 'bound': 'mfma', 'achieved': achieved, 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
            'frac': achieved / PEAK_F32_MFMA_TFLOPS, 'traffic': None,
            'launches': k['calls'], 'avg_launch_us': 1e3 * k['total_ms'] / k['calls'],
            'flops_per_launch_avg': flops / k['calls']}
