import numpy as np
import torch


def g(a, dev, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(dev)


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def max_abs(a, b):
    return float(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)).max())


def check_close(name, got, ref, rtol_l2=2e-5, atol=None):
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    assert got.shape == tuple(np.asarray(ref).shape), "%s: shape %s vs %s" % (name, got.shape, np.asarray(ref).shape)
    assert np.isfinite(got).all(), "%s: non-finite values" % name
    r, m = rel_l2(got, ref), max_abs(got, ref)
    ok = r <= rtol_l2 or (atol is not None and m <= atol)
    assert ok, "%s: rel-L2 %.3e (tol %.1e), max-abs %.3e (atol %s), ref-norm %.3e" % (name, r, rtol_l2, m, atol, np.linalg.norm(ref))
    return r, m
