"""Generates tests/golden/*.npz -- run once in the build container, outputs committed.

    python tests/golden/make_golden.py

Sources of the expected values
  * `known_answers`: the three values the reference itself publishes
    (CuSMC/CuSMC.tex:95-105, :131-142; man/metropolis_hastings.Rd:22-27).
  * `pdf_*`: scipy.stats.multivariate_normal / multivariate_t (scipy 1.15.3), an implementation
    independent of both the oracle and the HIP kernels, plus the oracle's reference-faithful
    restatement (per-particle LU determinant + inverse, src/statistics.cc.cpp:171-196,295-324).
    The reference has no fixtures of its own beyond the three known answers and
    cannot be built here, so these are "parity unpinned" against the reference binary.
  * `resample_*`, `pf_*`: the oracle under the build's Philox contract (the reference's RNG
    cannot be seeded: src/samplers.cpp:10-11).  They pin the contract against drift.
  * `pf_y`: the first 10 observation rows of the reference's example data data_raw/y_t.csv
    (an input data file, read as text here; /root/reference is not needed at test time).
"""
import os
import sys

import numpy as np
from scipy import stats

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import oracle as O  # noqa: E402


def spd(rng, d):
    A = rng.standard_normal((d, d))
    return A @ A.T / d + np.eye(d)


def main():
    out = {}
    # ---- density cases ---------------------------------------------------------------------
    dims = [1, 2, 3, 8, 32, 64, 65, 256]
    nus = {1: 3.0, 2: 4.0, 3: 2.5, 8: 30.0, 32: 4.0, 64: 4.0, 65: 2.5, 256: 4.0}
    for d in dims:
        rng = np.random.default_rng(1000 + d)
        N = 8 if d == 256 else (24 if d >= 64 else 40)
        sigma = spd(rng, d)
        mu = rng.standard_normal(d)
        X = mu + rng.standard_normal((N, d)) @ np.linalg.cholesky(sigma).T * 1.3
        F = np.eye(d) + 0.1 * rng.standard_normal((d, d)) / np.sqrt(d)
        y = F @ mu + 0.5 * rng.standard_normal(d)  # keeps y - F x moderate: densities stay > 0
        nu = float(np.float32(nus[d]))
        pre = "pdf_d%d_" % d
        out[pre + "X"], out[pre + "mu"], out[pre + "sigma"] = X, mu, sigma
        out[pre + "F"], out[pre + "y"], out[pre + "nu"] = F, y, np.float64(nu)
        # pdf(y, F): r = x - F mu
        out[pre + "mvn_logpdf_scipy"] = stats.multivariate_normal(F @ mu, sigma).logpdf(X).reshape(N)
        out[pre + "mvt_logpdf_scipy"] = stats.multivariate_t(F @ mu, sigma, df=nu).logpdf(X).reshape(N)
        out[pre + "mvn_pdf_oracle"] = O.pdf_batch(X, mu, sigma, F, "mvn")
        out[pre + "mvt_pdf_oracle"] = O.pdf_batch(X, mu, sigma, F, "mvt", nu)
        # reweight_G: w_i = pdf_{0,sigma}(y - F x_i)
        R = y - X @ F.T
        out[pre + "mvn_reweight_scipy"] = stats.multivariate_normal(np.zeros(d), sigma).logpdf(R).reshape(N)
        out[pre + "mvt_reweight_scipy"] = stats.multivariate_t(np.zeros(d), sigma, df=nu).logpdf(R).reshape(N)
        out[pre + "mvn_reweight_oracle"] = O.reweight(X, y, F, sigma, "mvn")
        out[pre + "mvt_reweight_oracle"] = O.reweight(X, y, F, sigma, "mvt", nu)
    # ---- resampler cases -------------------------------------------------------------------
    rng = np.random.default_rng(7)
    cases = {
        "equal": np.full(64, 0.25),
        "zeros": np.zeros(2),                      # the man-page example: NaN ratio never accepts
        "dominant": np.r_[np.full(99, 1e-12), 1.0],
        "negative": rng.standard_normal(100),      # the paper's rnorm(100) example
        "tiny": np.exp(-50 - 5 * rng.random(257)),  # ~1e-22..1e-24: exercises the ratio form
        "with_nan": np.r_[rng.random(31), np.nan],
    }
    for name, w in cases.items():
        for B in (1, 10, 37):
            out["resample_%s_w" % name] = w
            out["resample_%s_B%d" % (name, B)] = O.metropolis(w, B, seed=20240 + B, step=1)
    # ---- a small filter trajectory ----------------------------------------------------------
    y_rows = np.array([[0, 0], [-1.94145, -1.57094], [-2.07943, -1.67384], [-2.15117, -1.74655]])
    ref_csv = "/root/reference/data_raw/y_t.csv"
    if os.path.exists(ref_csv):
        y_rows = np.loadtxt(ref_csv, delimiter=",", skiprows=1)[:10]
    out["pf_y"] = y_rows
    d, N = 2, 64
    I = np.eye(d)
    for dist, nu in (("mvn", 0.0), ("mvt", 5.0)):
        X, w, a = O.pf_run(y_rows, N, np.zeros(d), I, I, I, 0.5 * I, 0.1 * I, dist, nu, B=10,
                           scale=1.0, seed=99)
        out["pf_%s_X" % dist], out["pf_%s_w" % dist], out["pf_%s_a" % dist] = X, w, a
    np.savez_compressed(os.path.join(HERE, "golden.npz"), **out)
    print("wrote", os.path.join(HERE, "golden.npz"), "with", len(out), "arrays")


if __name__ == "__main__":
    main()
