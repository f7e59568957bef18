"""Welch's unequal-variance comparison of two small samples of a metric - how the accuracy tests decide "indistinguishable" (test infrastructure).

Two pipelines (the product on the GPU, the oracle's SLAM loop on the CPU) produce one value of ATE RMSE / depth L1 per 50-frame run; the runs
are chaotic, so single runs scatter by 10-30 % in BOTH pipelines.  What CAN be stated is an interval for the difference of the means:

    diff = mean_a - mean_b,  se = sqrt(s_a^2 / n_a + s_b^2 / n_b),  dof by Welch-Satterthwaite,  half-width = t(1 - alpha / 2, dof) se

`resolvable_rel` = the 95 % half-width relative to mean_b: the smallest relative difference of the means these samples can tell from
zero - the honest replacement for "within 1 % of the reference" (BASELINE.json north_star) when the data cannot resolve 1 %."""
import math

import numpy as np
from scipy import stats


def welch(a, b, alpha=0.05):
    a, b = np.asarray(a, float), np.asarray(b, float)
    na, nb = a.size, b.size
    assert na >= 2 and nb >= 2, 'a variance needs two runs per side'
    va, vb = a.var(ddof=1) / na, b.var(ddof=1) / nb
    se = math.sqrt(va + vb)
    dof = (va + vb) ** 2 / (va ** 2 / (na - 1) + vb ** 2 / (nb - 1)) if se > 0 else float(na + nb - 2)
    diff = float(a.mean() - b.mean())
    t = float(stats.t.ppf(1 - alpha / 2, dof))
    p = float(2 * stats.t.sf(abs(diff) / se, dof)) if se > 0 else (1.0 if diff == 0 else 0.0)
    return dict(mean_a=float(a.mean()), mean_b=float(b.mean()), sd_a=float(a.std(ddof=1)), sd_b=float(b.std(ddof=1)), n_a=int(na), n_b=int(nb),
                diff=diff, rel_diff=diff / float(b.mean()), se=se, dof=float(dof), alpha=alpha, half_width=t * se,
                resolvable_rel=t * se / abs(float(b.mean())), p_value=p)


def indistinguishable(a, b, alpha_test=0.002):
    """(ok, record): the difference of the means lies inside its own 1 - alpha_test Welch interval (alpha_test 0.002: an honest pair of
    pipelines fails one comparison in 500 - the suite makes eight of them per run); the record carries the 95 % interval for the reports."""
    w95, wt = welch(a, b, 0.05), welch(a, b, alpha_test)
    rec = dict(w95, half_width_test=wt['half_width'], alpha_test=alpha_test)
    return abs(wt['diff']) <= wt['half_width'], rec
