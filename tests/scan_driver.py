"""Test-side restatement of ldb_amgettuple's paging loop (lantern_hnsw/src/hnsw/scan.c:167-338).

It drives any `search(k) -> (labels, dists, slots)` callable (a plain top-k search) the way the
PostgreSQL executor drives the index AM, so the same golden cases run against the oracle.  The
streaming continuation (usearch_search_ef(..., streaming=true), scan.c:273-281) is restated as the
device library implements it: search for |already returned| + k results and hand out the first k
that were not returned before.  (The device library has its own C++ scan shim,
lantern_amd/csrc/scan_shim.cpp, tested in tests/test_gpu_parity.py.)
"""
INVALID_ELEMENT_LABEL = 0  # lantern_hnsw/src/hnsw.h:40


class Scan:
    def __init__(self, search, index_size, init_k=10):
        self.search, self.index_size, self.init_k = search, index_size, init_k
        self.first = True
        self.labels, self.count, self.current = [], 0, 0
        self.seen = set()
        self.k_trace = []  # "LANTERN querying index for %d elements" (scan.c:219, :272)

    def _next_batch(self, k, streaming):
        self.k_trace.append(k)
        if not streaming:
            self.seen.clear()
        want = min(len(self.seen) + k, self.index_size)
        labels, _, slots = self.search(want)
        out = []
        for label, slot in zip(labels, slots):
            if len(out) == k:
                break
            if int(slot) in self.seen:
                continue
            self.seen.add(int(slot))
            out.append(int(label))
        return out

    def gettuple(self):
        """One ldb_amgettuple call: the next label, or None when the scan is exhausted."""
        if self.first:  # scan.c:181-238: k = lantern_hnsw.init_k, streaming=false
            self.labels = self._next_batch(self.init_k, False)
            self.count, self.current, self.first = len(self.labels), 0, False
        if self.current == self.count:  # scan.c:240-292
            k = self.count * 2
            if self.count >= 1000:  # scan.c:249-252 hard stop
                return None
            if self.index_size == self.current:  # scan.c:254-256
                return None
            if k == 0:
                return None
            self.labels = self._next_batch(k, True)  # streaming=true: the NEXT k
            self.count, self.current = len(self.labels), 0
        while self.current < self.count:  # scan.c:294-335
            label = int(self.labels[self.current])
            self.current += 1
            if label == INVALID_ELEMENT_LABEL:
                continue
            return label
        return None


def scan(search, index_size, limit, init_k=10, trace=None):
    """`limit` rows the way the executor pulls them; `trace` (a list) receives the k of every search the scan issued."""
    s, out = Scan(search, index_size, init_k), []
    while len(out) < limit:
        label = s.gettuple()
        if label is None:
            break
        out.append(label)
    if trace is not None:
        trace.extend(s.k_trace)
    return out
