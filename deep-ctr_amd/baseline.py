"""Host-side pieces of the reference's TensorFlow driver python/baseline.py that sit around the IP
family's train / eval calls: the smoothed-window early stop (:262-281) and the
negative-down-sampling re-calibration of predictions (:369, :422).  Same names and module-level knobs
as the driver; the Criteo ETL, buffered TSV reader and hard-coded algo table of that file are out of
scope.

Both are held, decision for decision, to runs of the reference's own statements
(tests/golden/ref_run.npz `es_*`, `nds_*`; tests/test_oracle_vs_reference.py)."""
import numpy as np

nds_rate = 0.025                   # python/baseline.py:21
least_step = 0                     # :169 sets 10 * epoch once the algo table has run; set before use
skip_window = 1                    # :59-61 are 1 / 10 / 10
smooth_window = 1
stop_window = 2


def re_calibrate(preds, rate=None):
    """p / (p + (1 - p) / nds_rate): predictions of a model trained on negatively down-sampled data back
    on the original scale (python/baseline.py:369, :422).  Returns a new array."""
    r = nds_rate if rate is None else rate
    p = np.asarray(preds, dtype=np.float64)
    return p / (p + (1 - p) / r)


def _window_mean(m, j):
    """Mean of m[j : j + smooth_window].  The terms are added in the order the driver's vector code adds them -- the
    window's LAST entry first, then the others front to back -- so the value (and with it the sign of a difference of
    two means of nearly equal windows, which is the decision) is the driver's to the bit; a running cumulative sum would be
    cheaper still but rounds differently, and a flat metric would then stop or not by round-off."""
    acc = m[j + smooth_window - 1]
    for v in m[j:j + smooth_window - 1]:
        acc = acc + v
    return acc / smooth_window


def smoothed_change(errs):
    """The quantity the stop rule looks at: every skip_window-th recorded metric, averaged over smooth_window entries; the
    LATEST average minus the one stop_window - 1 averages earlier.  None while there are fewer than stop_window averages."""
    m = [float(v) for v in errs[::skip_window]]
    n_avg = len(m) - smooth_window + 1
    if n_avg < stop_window or n_avg < 1:
        return None
    return _window_mean(m, n_avg - 1) - _window_mean(m, n_avg - stop_window)


def early_stop(step, errs, metric='auc'):
    """python/baseline.py:262-281.  Stop once the smoothed metric has moved the wrong way over the stop window (rmse up,
    auc down); never before `least_step`.  Only the two window means the decision needs are formed."""
    if step <= least_step:
        return False
    change = smoothed_change(errs)
    if change is None:
        return False
    worse = change > 0 if metric == 'rmse' else change < 0 if metric == 'auc' else False
    if worse:
        print('early stop at step %d' % step)
        print('smoothed %s error %g' % (metric, change))
    return bool(worse)


def keep_recent(errs):
    """The driver keeps only as many recorded metrics as the rule can look at (python/baseline.py:373)."""
    return errs[-2 * skip_window * (stop_window + smooth_window):]
