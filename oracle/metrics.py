"""Oracle SDR / SI-SDR (TEST INFRASTRUCTURE - see oracle/__init__.py).  Restates /root/reference/utils.py:148-200."""
import numpy as np


def calculate_sdr(ref: np.ndarray, est: np.ndarray, eps=1e-10) -> float:
    """utils.py:148-169."""
    noise = est - ref
    numerator = np.clip(a=np.mean(ref ** 2), a_min=eps, a_max=None)
    denominator = np.clip(a=np.mean(noise ** 2), a_min=eps, a_max=None)
    return 10.0 * np.log10(numerator / denominator)


def calculate_sisdr(ref: np.ndarray, est: np.ndarray) -> float:
    """utils.py:172-200 - everything in the input dtype; eps = finfo(ref.dtype).eps."""
    eps = np.finfo(ref.dtype).eps
    reference = ref.copy().reshape(ref.size, 1)
    estimate = est.copy().reshape(est.size, 1)
    rss = np.dot(reference.T, reference)
    a = (eps + np.dot(reference.T, estimate)) / (rss + eps)
    e_true = a * reference
    e_res = estimate - e_true
    sss = (e_true ** 2).sum()
    snn = (e_res ** 2).sum()
    return 10 * np.log10((eps + sss) / (eps + snn))
