"""Synthetic vector sets of the benchmark: numpy default_rng, f32, C-contiguous.  Shared by bench.py,
the tests and the work-sharded build child (lantern_amd/sharded_build.py) so that all build from the very same rows.

  gaussian   the prescribed i.i.d. N(0,1) set.  In 768 dimensions it has no neighbourhood structure: distances
             concentrate, recall@10 of ANY HNSW at M=16 / ef=64 is ~0.18 (CPU port and device agree exactly).
  lowrank    32 latent dims embedded in `dim` + 5 % isotropic noise: recall@10 ~0.64.
  clustered  a Gaussian mixture with low-dimensional clusters -- the regime of real embedding sets, and the one in which
             the reference asserts recall (>= 0.7 hard floor, warning below 0.9: scripts/integration_tests.py:249-264):
             16 centres ~ N(0, I_dim); a point = centre_c + (z @ P) * s_c + 0.05 * N(0, I_dim), z ~ N(0, I_8), P a fixed
             8 x dim projection scaled by 1/sqrt(8), s_c a fixed +-1 sign pattern per cluster (every cluster spans its own
             8-dimensional subspace).  Queries are drawn from the same mixture.  recall@10 at M=16, ef_construction=128,
             ef=64 on 1M x 768 rows: 0.955 on the CPU port (the device's batch plan); profiles/ has the device's figure.
             (What decides recall here is how many separate clusters a walk has to find its way between, more than their
             intrinsic dimension: 64 clusters of 12 latent dimensions give 0.87 at 1M rows, 64 of 8 give 0.83, 512 give 0.77.)
"""
from __future__ import annotations

import numpy as np

BASE_SEED = 3
CLUSTERS, CLUSTER_LATENT, CLUSTER_NOISE, CLUSTER_SEED, CLUSTER_CHUNK = 16, 8, 0.05, 77, 65536
CLUSTERED_DOC = (f"{CLUSTERS} Gaussian clusters, each in its own {CLUSTER_LATENT}-dimensional subspace + {CLUSTER_NOISE} isotropic noise, "
                 f"structure seed {CLUSTER_SEED}")


def query_maker(kind: str, dim: int):
    """rows(rng, n) for the named data kind (see the module docstring)."""
    if kind == "gaussian":
        return lambda r, n: r.standard_normal((n, dim), dtype=np.float32)
    if kind == "lowrank":
        proj = np.random.default_rng(33).standard_normal((32, dim), dtype=np.float32) / np.float32(np.sqrt(32))

        def make(r, n):
            z = r.standard_normal((n, 32), dtype=np.float32)
            return z @ proj + np.float32(0.05) * r.standard_normal((n, dim), dtype=np.float32)

        return make
    if kind == "clustered":
        s = np.random.default_rng(CLUSTER_SEED)
        centres = s.standard_normal((CLUSTERS, dim), dtype=np.float32)
        proj = s.standard_normal((CLUSTER_LATENT, dim), dtype=np.float32) / np.float32(np.sqrt(CLUSTER_LATENT))
        signs = np.where(s.standard_normal((CLUSTERS, dim), dtype=np.float32) < 0, np.float32(-1), np.float32(1))

        def make(r, n):
            # fixed-size chunks, each drawing (cluster ids, latent coordinates, noise) in that order: the first rows of a
            # larger set are the rows of a smaller one (up to the last partial chunk), and the temporaries stay small
            out = np.empty((n, dim), dtype=np.float32)
            for lo in range(0, n, CLUSTER_CHUNK):
                m = min(CLUSTER_CHUNK, n - lo)
                c = r.integers(0, CLUSTERS, m)
                z = r.standard_normal((m, CLUSTER_LATENT), dtype=np.float32)
                x = (z @ proj) * signs[c]
                x += centres[c]
                x += np.float32(CLUSTER_NOISE) * r.standard_normal((m, dim), dtype=np.float32)
                out[lo:lo + m] = x
            return out

        return make
    raise ValueError(f"unknown data kind {kind!r}")


def base_rows(kind: str, n: int, dim: int, seed: int = BASE_SEED) -> np.ndarray:
    return query_maker(kind, dim)(np.random.default_rng(seed), n)


def shard_rows(kind: str, n: int, dim: int, lo: int, hi: int) -> np.ndarray:
    """Rows [lo, hi) of a multi-rank job's base set WITHOUT generating the rest: every rank of `bench.py --gpus N` draws its own
    shard from its own stream (seed = (BASE_SEED, lo)), so host memory and generation time per rank fall with the world size
    instead of every rank synthesising all n rows.  The set is the same distribution as base_rows(kind, n, dim), not the same
    rows (a rank cannot skip ahead in a Gaussian stream); a one-rank job keeps base_rows."""
    return query_maker(kind, dim)(np.random.default_rng([BASE_SEED, lo]), hi - lo)
