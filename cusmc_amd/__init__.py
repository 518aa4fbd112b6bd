"""cusmc_amd -- MI355X (gfx950) implementation of CuSMC's per-particle likelihood / proposal /
accept-reject hot path, behind the reference's own operator names.

  csrc/      hand-written HIP kernels + the C ABI (include/cusmc_hip.h) -> libcusmc_hip.so
  api.py     host-side mirror of the reference interface (MVNPDF, MVTPDF, MVN, MVT,
             metropolis_hastings, run; MultiVariate*Distribution, Sampler)
  sharding.py  one-process-per-GPU partitioning of particles / chains (torch.distributed)
  io.py      writeOutput() CSV side effects of run()

The compute lives in the shared library only; nothing here falls back to numpy or torch math.
"""
from ._lib import SO_PATH, CusmcError  # noqa: F401
from .api import (MVN, MVNPDF, MVT, MVTPDF, Context, MultiVariateNormalDistribution,  # noqa: F401
                  MultiVariateTStudentDistribution, Sampler, cholesky_batched, eigenSolver,
                  logpdf_percov, metropolis_hastings, run, set_seed)
