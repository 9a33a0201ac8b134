import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from test_gpu_ops import dense_gpu, rng, O, _gemm_class_launches
from vaenar_tts_amd import _lib
from vaenar_tts_amd.configs import tiny_hps
eng = _lib.Engine(tiny_hps(), 0)
for rows in ("window", "wide"):
  for (m,k1,k2,n,ln) in [(200, 512, 0, 256, 0), (64, 100, 0, 513, 0), (333, 256, 256, 256, 1)]:
    r = rng(m + n + k1)
    mags = [0.05, 1.0, 30.0, 300.0] if rows == "window" else [1e-5, 1e-3, 1.0, 30.0, 1e5]
    sc = r.choice(mags, size=(m, 1))
    a1 = r.standard_normal((m, k1)) * sc
    a2 = (r.standard_normal((m, k2)) * (np.abs(a1).max(1, keepdims=True) / 4)) if k2 else None
    w = r.standard_normal((k1 + k2, n)) / np.sqrt(k1 + k2)
    b = r.standard_normal(n) * (0.0 if not ln else 1.0)
    res = r.standard_normal((m, n)) if ln else None
    g, be = 1 + 0.1 * r.standard_normal(n), 0.1 * r.standard_normal(n)
    f = lambda x: np.asarray(x, np.float32).astype(np.float64)
    eng.set_option("op_dense_split", 1)
    got, ns, ne = _gemm_class_launches(eng, lambda: dense_gpu(eng, a1, w, a2=a2, bias=b, residual=res, ln=(g, be) if ln else None))
    eng.set_option("op_dense_split", 0)
    x = f(a1) if a2 is None else np.concatenate([f(a1), f(a2)], -1)
    ref = O.dense(x, f(w), f(b))
    if ln: ref = O.layer_norm(f(res) + ref, f(g), f(be))
    pscale = np.sqrt((x * x).sum(1, keepdims=True) / (k1 + k2))
    rel = np.abs(got - ref) / (pscale if not ln else 1.0)
    print(rows, (m,k1,k2,n,ln), "launches split/exact", ns, ne, "finite", np.isfinite(got).all())
    for mg in mags:
        sel = (sc[:,0] == mg)
        if sel.any(): print("   rows of magnitude %-8g: worst err / row product scale %.3e" % (mg, rel[sel].max()))
