"""Host-side pieces of the reference's TensorFlow driver python/baseline.py that sit around the IP
family's train / eval calls: the smoothed-window early stop (:262-281) and the
negative-down-sampling re-calibration of predictions (:368-369, :422).  Plain NumPy, same names and
module-level knobs; the Criteo ETL, buffered TSV reader and hard-coded algo table of that file are
out of scope."""
import numpy as np

nds_rate = 0.025                   # python/baseline.py:21
least_step = 0                     # :31-35 (defaults of the reference's header block; set before use)
skip_window = 1
smooth_window = 1
stop_window = 2


def re_calibrate(preds, rate=None):
    """p / (p + (1 - p) / nds_rate): predictions of a model trained on negatively down-sampled data back
    on the original scale (python/baseline.py:368-369, :422).  Returns a new array."""
    r = nds_rate if rate is None else rate
    p = np.asarray(preds, dtype=np.float64)
    return p / (p + (1 - p) / r)


def early_stop(step, errs, metric='auc'):
    """python/baseline.py:262-281: every skip_window-th entry of `errs`, a moving average over
    smooth_window entries, then the change over stop_window - 1 smoothed points; stop when the latest
    change goes the wrong way (rmse up, auc down).  Never before `least_step`."""
    if step > least_step:
        skip_metric = np.asarray(errs[::skip_window], dtype=np.float64)
        smooth_metric = np.array(skip_metric[smooth_window - 1:])
        for i in range(smooth_window - 1):
            smooth_metric += skip_metric[i:(i - smooth_window + 1)]
        smooth_metric /= smooth_window
        if len(smooth_metric) < stop_window:
            return False
        smooth_error = smooth_metric[stop_window - 1:] - smooth_metric[:1 - stop_window]
        if metric == 'rmse' and smooth_error[-1] > 0:
            print('early stop at step %d' % step)
            print('smoothed rmse error', str(smooth_error))
            return True
        elif metric == 'auc' and smooth_error[-1] < 0:
            print('early stop at step %d' % step)
            print('smoothed auc error', str(smooth_error))
            return True
        return False
    return False
