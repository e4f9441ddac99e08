"""A stand-in for the PostgreSQL side of a Lantern index, for tests: node tapes in 8 KB pages addressed by 6-byte
ItemPointers, the retriever callbacks, and the slot rewrite StoreExternalIndex performs on import.

Follows lantern_hnsw/src/hnsw/external_index.c:46-177 (StoreExternalIndexNodes: how node tapes are packed into pages),
:240-432 (StoreExternalIndex: header page, codebook pages of a pq index, node i goes to some (block, offset); every
neighbour slot -- a u32 sequential id in the low 4 of its 6 bytes -- and the header's entry slot are rewritten to that
ItemPointer), :613-697 (retriever / retriever_mut: slot -> pointer to the tape) and usearch_storage.cpp:19-44
(tape layout; usearch_init_node zeroes a new tape and sets key + level only).

The page arithmetic is PostgreSQL's (bufpage.h / bufpage.c, BLCKSZ 8192): a page starts with a 24-byte PageHeaderData,
line pointers (ItemIdData, 4 bytes each) grow up from it, items grow down from the special area (here
MAXALIGN(sizeof(HnswIndexPageSpecialBlock)) = 16 bytes, external_index.h:68-74), every item is stored MAXALIGNed (8).
An item is an HnswIndexTuple: seqid u32, size u32, then the node tape (external_index.h:76-82).
"""
from __future__ import annotations

import ctypes as C
import struct

import numpy as np

BLCKSZ = 8192
PAGE_HEADER = 24          # SizeOfPageHeaderData
ITEM_ID = 4               # sizeof(ItemIdData)
SPECIAL = 16              # MAXALIGN(sizeof(HnswIndexPageSpecialBlock)): three uint32 -> 12 -> 16
TUPLE_HEADER = 8          # offsetof(HnswIndexTuple, node): seqid u32 + size u32


def maxalign(n: int) -> int:
    return (n + 7) & ~7


def pack_nodes(node_sizes, first_block: int = 1):
    """StoreExternalIndexNodes (external_index.c:46-177) over nodes of the given tape sizes, in order, starting on a fresh
    page at `first_block`: a new page is begun when PageGetFreeSpace(page) < sizeof(HnswIndexTuple) + node_size (:104) or
    when PageAddItem refuses the MAXALIGNed item (:151-155).  Returns [(block, offset)] per node and the number of data
    pages used."""
    out = []
    block = first_block - 1
    lower = upper = 0
    fresh = True
    for size in node_sizes:
        item = TUPLE_HEADER + size
        assert PAGE_HEADER + ITEM_ID + maxalign(item) + SPECIAL <= BLCKSZ, "node does not fit a page (external_index.c:60)"
        while True:
            free = max(0, upper - lower - ITEM_ID) if not fresh else -1    # PageGetFreeSpace: room left after one more line pointer
            if fresh or free < item:
                block += 1
                lower, upper, fresh = PAGE_HEADER, BLCKSZ - SPECIAL, False
                continue
            if lower + ITEM_ID > upper - maxalign(item):                   # PageAddItem: InvalidOffsetNumber -> force a new page
                fresh = True
                continue
            lower += ITEM_ID
            upper -= maxalign(item)
            out.append((block, (lower - PAGE_HEADER) // ITEM_ID))
            break
    return out, (block - first_block + 1 if out else 0)


def index_relation_pages(node_sizes, pq: bool = False, dims: int = 0) -> int:
    """Pages of the whole index relation StoreExternalIndex leaves behind (external_index.c:240-432): the header page, for a
    pq index ceil(256 * dims * 4 / BLCKSZ) codebook pages (:283-296), then the data pages."""
    codebook = -(-256 * dims * 4 // BLCKSZ) if pq else 0
    return 1 + codebook + pack_nodes(node_sizes, 1 + codebook)[1]


def item_pointer_of(block: int, pos: int) -> int:
    """ItemPointerData{bi_hi u16, bi_lo u16, posid u16} as the low 48 bits of a u64."""
    return int.from_bytes(struct.pack("<HHH", block >> 16, block & 0xFFFF, pos), "little")


class PageStore:
    def __init__(self, capi, blob: bytes, dims_bytes: int, M: int):
        """blob: a usearch-format file (136-byte header + tapes with sequential-id slots)."""
        self.capi, self.M, self.vb = capi, M, dims_bytes
        self.pages: dict[int, C.Array] = {}
        self.order: list[int] = []  # slot of the i-th node (sequential id -> page slot)
        self.retrieved: list[int] = []
        self.mutated: list[int] = []
        blob = bytearray(blob)
        tapes, off = [], 136
        while off < len(blob):
            level = struct.unpack_from("<H", blob, off + 8)[0]
            size = self.tape_bytes(level)
            tapes.append(bytearray(blob[off:off + size]))
            off += size
        assert off == len(blob)
        self.placed, self.data_pages = pack_nodes([len(t) for t in tapes])
        item_pointer = lambda i: item_pointer_of(*self.placed[i])
        for i, t in enumerate(tapes):
            for l, q, cap, cnt in self._lists(t):
                for j in range(cnt):
                    seq = struct.unpack_from("<I", t, q + 4 + j * 6)[0]
                    t[q + 4 + j * 6:q + 10 + j * 6] = item_pointer(seq).to_bytes(6, "little")
            slot = item_pointer(i)
            self.pages[slot] = C.create_string_buffer(bytes(t), len(t))
            self.order.append(slot)
        hbuf = C.create_string_buffer(bytes(blob[:136]), 136)
        if tapes:
            entry_seq = capi.lib().usearch_header_get_entry_slot(hbuf)
            capi.lib().usearch_header_set_entry_slot(hbuf, item_pointer(entry_seq))
        self.header = hbuf.raw[:136]

    def tape_bytes(self, level: int) -> int:
        return 10 + (4 + 2 * self.M * 6) + level * (4 + self.M * 6) + self.vb

    def _lists(self, tape):
        level = struct.unpack_from("<H", tape, 8)[0]
        q = 10
        for l in range(level + 1):
            cap = 2 * self.M if l == 0 else self.M
            yield l, q, cap, struct.unpack_from("<I", tape, q)[0]
            q += 4 + cap * 6

    # ---- the callbacks usearch gets (external_index.c:613-697) ------------------------------------------------------
    def retriever(self, slot: int) -> int:
        self.retrieved.append(slot)
        return C.addressof(self.pages[slot])

    def retriever_mut(self, slot: int) -> int:
        self.mutated.append(slot)
        return C.addressof(self.pages[slot])

    # ---- what ldb_aminsert does around usearch_add_external (insert.c:182-214) ------------------------------------
    def new_tuple(self, label: int, level: int) -> tuple[int, int]:
        """PrepareIndexTuple + usearch_init_node: a zeroed tape with key and level set.  Returns (address, slot).
        (external_index.c:478-560: the tuple goes to the last data page if it fits, else to a new one.)"""
        sizes = [len(self.pages[s]) for s in self.order] + [self.tape_bytes(level)]
        self.placed, self.data_pages = pack_nodes(sizes)
        slot = item_pointer_of(*self.placed[-1])
        buf = C.create_string_buffer(self.tape_bytes(level))
        struct.pack_into("<QH", buf, 0, label, level)
        self.pages[slot] = buf
        self.order.append(slot)
        return C.addressof(buf), slot

    # ---- reading the pages back ----------------------------------------------------------------------------------
    def node(self, slot: int):
        """(label, level, [neighbour slots per level], vector bytes) of the node at `slot`."""
        t = bytes(self.pages[slot].raw)
        label, level = struct.unpack_from("<QH", t, 0)
        lists = []
        q = 10
        for l, q, cap, cnt in self._lists(t):
            lists.append([int.from_bytes(t[q + 4 + j * 6:q + 10 + j * 6], "little") for j in range(cnt)])
            # unused slots stay zero (validate_index.c:140-151)
            assert t[q + 4 + cnt * 6:q + 4 + cap * 6] == bytes((cap - cnt) * 6)
        vec_off = 10 + (4 + 2 * self.M * 6) + level * (4 + self.M * 6)
        return label, level, lists, t[vec_off:vec_off + self.vb]

    def graph_by_label(self):
        """{label: (level, [[neighbour labels] per level], vector bytes)} -- comparable across differently numbered indexes."""
        lab = {s: struct.unpack_from("<Q", self.pages[s].raw, 0)[0] for s in self.order}
        out = {}
        for s in self.order:
            label, level, lists, vec = self.node(s)
            out[label] = (level, [[lab[x] for x in lst] for lst in lists], vec)
        return out


def graph_by_label(g, vectors: np.ndarray):
    """The same view of an exported device / oracle graph (capi.GpuIndex.export_graph)."""
    M0 = g["nbr0"].shape[1]
    M = M0 // 2
    labels = g["labels"]
    out = {}
    for i in range(len(labels)):
        lists = [[int(labels[x]) for x in g["nbr0"][i] if x != 0xFFFFFFFF]]
        for l in range(1, int(g["levels"][i]) + 1):
            row = g["upper_nbr"][int(g["upper_off"][i]) + l - 1]
            lists.append([int(labels[x]) for x in row if x != 0xFFFFFFFF])
        out[int(labels[i])] = (int(g["levels"][i]), lists, np.ascontiguousarray(vectors[i]).tobytes())
    return out
