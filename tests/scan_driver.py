"""Test-side restatement of ldb_amgettuple's paging loop (lantern_hnsw/src/hnsw/scan.c:167-338).

It drives any `search(k, skip) -> (labels, dists)` callable the way the PostgreSQL executor
drives the index AM, so the same golden cases run against the oracle.  (The device library
has its own C++ scan shim, lantern_amd/csrc/scan_shim.cpp, tested in test_gpu_scan.py.)
"""
INVALID_ELEMENT_LABEL = 0  # lantern_hnsw/src/hnsw.h:40


class Scan:
    def __init__(self, search, index_size, init_k=10):
        self.search, self.index_size, self.init_k = search, index_size, init_k
        self.first = True
        self.labels, self.count, self.current, self.returned_total = [], 0, 0, 0

    def gettuple(self):
        """One ldb_amgettuple call: the next label, or None when the scan is exhausted."""
        if self.first:  # scan.c:181-238: k = lantern_hnsw.init_k, streaming=false
            self.labels, _ = self.search(self.init_k, 0)
            self.count, self.current, self.first = len(self.labels), 0, False
        if self.current == self.count:  # scan.c:240-292
            k = self.count * 2
            if self.count >= 1000:  # scan.c:249-252 hard stop
                return None
            if self.index_size == self.current:  # scan.c:254-256
                return None
            self.returned_total += self.count
            self.labels, _ = self.search(k, self.returned_total)  # streaming=true: the NEXT k
            self.count, self.current = len(self.labels), 0
        while self.current < self.count:  # scan.c:294-335
            label = int(self.labels[self.current])
            self.current += 1
            if label == INVALID_ELEMENT_LABEL:
                continue
            return label
        return None


def scan(search, index_size, limit, init_k=10):
    s, out = Scan(search, index_size, init_k), []
    while len(out) < limit:
        label = s.gettuple()
        if label is None:
            break
        out.append(label)
    return out
