"""Synthetic vector sets of the benchmark: numpy default_rng, f32, C-contiguous.  Shared by bench.py
and the work-sharded build child (lantern_amd/sharded_build.py) so that both build from the very same rows."""
from __future__ import annotations

import numpy as np

BASE_SEED = 3


def query_maker(kind: str, dim: int):
    """rows(rng, n) for the named data kind.
    gaussian = the prescribed i.i.d. N(0,1) set; lowrank = 32 latent dims embedded in `dim` + 5 % isotropic noise
    (has neighbourhood structure, unlike i.i.d. N(0,1) in 768-d)."""
    if kind == "gaussian":
        return lambda r, n: r.standard_normal((n, dim), dtype=np.float32)
    proj = np.random.default_rng(33).standard_normal((32, dim), dtype=np.float32) / np.float32(np.sqrt(32))

    def make(r, n):
        z = r.standard_normal((n, 32), dtype=np.float32)
        return z @ proj + np.float32(0.05) * r.standard_normal((n, dim), dtype=np.float32)

    return make


def base_rows(kind: str, n: int, dim: int) -> np.ndarray:
    return query_maker(kind, dim)(np.random.default_rng(BASE_SEED), n)
