"""Golden vectors for the evaluation metrics, from the reference's own evaluation.py.

Runs ONLY in the build container (the reference checkout is mounted at /root/reference and needs scikit-learn);
writes data only -- label / score vectors and the numbers the reference's functions return for them -- to
tests/golden/g8_metrics.npz.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
import evaluation as ref   # noqa: E402


def main():
    rng = np.random.default_rng(1798)
    cases = {}

    def add(name, y, s):
        cases[name] = (np.asarray(y, dtype=np.int64), np.asarray(s, dtype=np.float64))

    n = 4000
    y = (rng.random(n) < 0.03).astype(np.int64)
    add("imbalanced", y, rng.normal(size=n) + 1.5 * y)
    add("ties", y, np.round(rng.normal(size=n) + 1.2 * y, 1))
    add("perfect", y, y * 5.0 + rng.random(n) * 0.1)
    add("inverted", y, -(rng.normal(size=n) + 1.5 * y))
    y2 = (rng.random(300) < 0.2).astype(np.int64)
    add("small", y2, rng.normal(size=300) + y2)
    add("coarse", y2, np.round(rng.normal(size=300) + y2))
    out = {}
    for name, (yy, ss) in cases.items():
        out[f"{name}/y"] = yy
        out[f"{name}/score"] = ss
        out[f"{name}/logauc"] = np.float64(ref.calculate_logAUC(yy, ss))
        out[f"{name}/logauc_wide"] = np.float64(ref.calculate_logAUC(yy, ss, FPR_range=(0.01, 0.5)))
        out[f"{name}/auc"] = np.float64(ref.calculate_auc(yy, ss))
        out[f"{name}/ppv"] = np.float64(ref.calculate_ppv(yy, ss))
        out[f"{name}/ppv_cut"] = np.float64(ref.calculate_ppv(yy, ss, cutoff=0.8))
        out[f"{name}/accuracy"] = np.float64(ref.calculate_accuracy(yy, ss))
        out[f"{name}/f1"] = np.float64(ref.calculate_f1_score(yy, ss))
    out["oneclass/auc"] = np.float64(ref.calculate_auc(np.zeros(10, dtype=np.int64), rng.normal(size=10)))
    path = os.path.join(HERE, "g8_metrics.npz")
    np.savez_compressed(path, **out)
    print(f"wrote g8_metrics.npz: {os.path.getsize(path) / 1024:.1f} KiB, {len(out)} arrays")
    for k in sorted(out):
        if "/y" not in k and "/score" not in k:
            print(k, float(out[k]))


if __name__ == "__main__":
    main()
