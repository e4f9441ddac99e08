"""File-based rendezvous for the ranks of ONE node (bench.py --gpus N, lantern_amd/sharded_build.py).

The data path of a multi-GPU run needs no host library at all (queries shard with no collective; the sharded build
exchanges over RCCL inside liblantern_gpu.so), so the ranks only have to agree on three small things: the RCCL unique
id, "everybody is here" (a barrier) and the max of a timing.  A directory under /dev/shm does that without importing
PyTorch into the measuring process (torch bundles a second HIP runtime; see lantern_amd/capi.py) and without a
listening socket.  The same object serves as the HOST transport of lantern_gpu_add_sharded
(capi.Comm.host(rank, world, rdv.allgatherv)) -- the fallback when RCCL cannot be brought up.

Every value is written to a temporary name and renamed into place, so a reader never sees a torn file.
"""
from __future__ import annotations

import os
import shutil
import time

import numpy as np


class RendezvousTimeout(TimeoutError):
    pass


def default_root() -> str:
    return "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else "/tmp"


def launch_key() -> str:
    """One key per launch: torch.distributed.run gives every worker the same MASTER_PORT and the same parent (the
    elastic agent), so a stale directory of an earlier launch on the same port is never picked up."""
    return f"lantern_rdv_{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}"


class FileRendezvous:
    def __init__(self, rank: int, world: int, path: str | None = None, timeout: float = 600.0, poll: float = 0.0005):
        self.rank, self.world, self.timeout, self.poll = rank, world, timeout, poll
        self.path = path or os.path.join(default_root(), launch_key())
        os.makedirs(self.path, exist_ok=True)
        self._seq = 0

    # ---- primitives ------------------------------------------------------------------------------------------
    def put(self, name: str, data: bytes):
        final = os.path.join(self.path, name)
        tmp = f"{final}.tmp{self.rank}"
        with open(tmp, "wb") as f:
            f.write(data)
        os.replace(tmp, final)

    def get(self, name: str, timeout: float | None = None) -> bytes:
        final = os.path.join(self.path, name)
        deadline = time.monotonic() + (self.timeout if timeout is None else timeout)
        while True:
            try:
                with open(final, "rb") as f:
                    return f.read()
            except FileNotFoundError:
                if time.monotonic() > deadline:
                    raise RendezvousTimeout(f"rank {self.rank}: nothing at {final} (a peer rank is missing)") from None
                time.sleep(self.poll)

    # ---- collectives -----------------------------------------------------------------------------------------
    # Every collective is one numbered all-gather of byte strings; calls must be made in the same order by every rank.
    # A rank that STARTS exchange s has completed s-1, i.e. has read every peer's s-1 file, i.e. every peer has started
    # s-1 and therefore finished reading the files of s-2: so each rank deletes its own s-2 file when it starts s, and
    # the directory never holds more than two exchanges (a sharded build over this transport moves gigabytes).
    def _exchange(self, data: bytes, timeout: float | None = None, skip=()) -> list:
        s = self._seq
        self._seq += 1
        if s >= 2:
            try:
                os.unlink(os.path.join(self.path, f"x{s - 2}.{self.rank}"))
            except FileNotFoundError:
                pass
        self.put(f"x{s}.{self.rank}", data)
        return [data if r == self.rank else (None if r in skip else self.get(f"x{s}.{r}", timeout)) for r in range(self.world)]

    def allgather(self, data: bytes, timeout: float | None = None) -> list[bytes]:
        """Every rank's `data`, in rank order."""
        return self._exchange(data, timeout)

    def barrier(self, timeout: float | None = None):
        self._exchange(b"", timeout)

    def broadcast(self, data: bytes | None, src: int = 0, timeout: float | None = None) -> bytes:
        return self._exchange((data or b"") if self.rank == src else b"", timeout)[src]

    def max_float(self, value: float, timeout: float | None = None) -> float:
        return max(float(np.frombuffer(b, dtype=np.float64)[0]) for b in self.allgather(np.float64(value).tobytes(), timeout))

    def allgatherv(self, buf: np.ndarray, offsets, counts):
        """In-place all-gather with per-rank sizes over a host buffer: the callback shape of capi.Comm.host."""
        o, c = offsets[self.rank], counts[self.rank]
        parts = self._exchange(buf[o:o + c].tobytes())
        for r in range(self.world):
            if r == self.rank or counts[r] == 0:
                continue
            if len(parts[r]) != counts[r]:
                raise RuntimeError(f"rank {self.rank}: segment of rank {r} has {len(parts[r])} bytes, expected {counts[r]}")
            buf[offsets[r]:offsets[r] + counts[r]] = np.frombuffer(parts[r], dtype=np.uint8)

    def finalize(self, timeout: float = 60.0):
        """The last call of every rank: a barrier, then rank 0 removes the directory -- only after every peer has said that
        it is through that barrier (a peer still polling for rank 0's file must not find the directory gone)."""
        self.barrier(timeout)
        self.put(f"done.{self.rank}", b"")
        if self.rank == 0:
            try:
                for r in range(1, self.world):
                    self.get(f"done.{r}", timeout)
            finally:
                shutil.rmtree(self.path, ignore_errors=True)
