"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/lantern_gpu.h
declares, and refuses loudly to compute without a device (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def capi():
    from lantern_amd import build, capi

    build.build()
    capi.lib()
    return capi


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "lantern_gpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = text.replace("#define LANTERN_GPU_EXPORT", "")
    names = re.findall(r"LANTERN_GPU_EXPORT[^;(]*?\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", text)
    assert len(names) > 30
    return sorted(set(names))


def test_library_exports_every_declared_symbol(capi):
    raw = C.CDLL(capi.LIB_PATH)
    missing = [n for n in declared_symbols() if not hasattr(raw, n)]
    assert not missing, f"declared in lantern_gpu.h but not exported: {missing}"


def test_binding_covers_the_header(capi):
    assert sorted(capi.EXPORTS) == declared_symbols()


def test_no_oracle_in_product_path():
    # the product must never import, load or link the oracle
    for dirpath, _, files in os.walk(os.path.join(ROOT, "lantern_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                for needle in ("import oracle", "from oracle", "liblantern_oracle", "lantern_oracle.h", "lo_search", "lo_distance"):
                    assert needle not in src, (f, needle)


def test_header_helpers_work_without_a_device(capi):
    buf = C.create_string_buffer(capi.USEARCH_HEADER_SIZE)
    capi.lib().usearch_header_set_entry_slot(buf, 0x0000123456789ABC)
    assert capi.lib().usearch_header_get_entry_slot(buf) == 0x0000123456789ABC


def test_fails_loudly_without_device(capi):
    if capi.device_count() > 0:
        pytest.skip("a device is present")
    with pytest.raises(capi.LanternGpuError, match="no HIP device"):
        capi.GpuIndex("l2sq", 8)
    with pytest.raises(capi.LanternGpuError, match="no HIP device"):
        capi.distance([1, 2, 3], [3, 2, 1], "l2sq")


def test_argument_validation_precedes_device_use(capi):
    # dimension mismatch text is the reference's (hnsw.c:301-303; hnsw_dist_func.out:126-135)
    with pytest.raises(capi.LanternGpuError, match="expected equally sized arrays but got arrays with dimensions 2 and 3"):
        capi.l2sq_dist([1, 1], [0, 1, 0])
    with pytest.raises(capi.LanternGpuError, match="expected equally sized arrays"):
        capi.hamming_dist([1, 1], [0, 1, 0])
    o = capi.InitOptions()
    o.metric_kind, o.quantization, o.dimensions, o.connectivity = 2, 1, 8, 16  # ip: not a Lantern metric
    err = C.c_char_p()
    assert capi.lib().usearch_init(C.byref(o), None, C.byref(err)) is None
    assert b"unsupported metric" in err.value
    o.metric_kind, o.pq = 3, True
    assert capi.lib().usearch_init(C.byref(o), None, C.byref(err)) is None
    assert b"pq = true needs a codebook" in err.value  # build.c:497-500 always passes the codebook it loaded
    o.pq, o.quantization, o.metric_kind = False, 4, 8  # hamming over i8 scalars: hamming is a metric over bits
    assert capi.lib().usearch_init(C.byref(o), None, C.byref(err)) is None
    assert b"hamming needs b1 scalars" in err.value
    o.quantization, o.metric_kind = 5, 1  # quant_bits = 1 on a cosine index is accepted since round 3: without a device it fails on THAT
    h = capi.lib().usearch_init(C.byref(o), None, C.byref(err))
    if capi.device_count() > 0:
        assert h is not None
        capi.lib().usearch_free(h, C.byref(err))
    else:
        assert h is None and b"no HIP device" in err.value


def test_level_draw_and_batch_plan_agree_with_the_oracle(capi):
    """The builder's two host-side rules (host_util.hpp) against the oracle's restatement (oracle/hnsw.c: lo_level_for,
    lo_plan_batch; level formula of lantern_hnsw/src/hnsw/insert.c:32-46): same level for every (seed, slot, M), same
    batch boundaries for every state -- the precondition of the edge-for-edge build parity the GPU tests assert."""
    import numpy as np

    from oracle import binding as oracle

    oracle.build()
    rng = np.random.default_rng(0)
    for M in (2, 3, 16, 48, 128):
        seeds = rng.integers(0, 2**63, 40)
        for seed in seeds[:4]:
            lv = [capi.level_for(int(seed), s, M) for s in range(3000)]
            assert lv == [oracle.level_for(int(seed), s, M) for s in range(3000)]
        big = np.array([capi.level_for(42, s, M) for s in range(60000)])
        assert abs((big >= 1).mean() - 1.0 / M) < 0.012  # P(level >= 1) = 1/M
        assert big.max() < 40
    for _ in range(300):
        size = int(rng.integers(0, 200000))
        max_level = int(rng.integers(0, 6))
        pending = rng.integers(0, max_level + 2, int(rng.integers(1, 600))).astype(np.int32)
        mb, mr = int(rng.integers(1, 9000)), int(rng.integers(1, 64))
        got = capi.plan_batch(size, max_level, pending, mb, mr)
        assert got == oracle.plan_batch(size, max_level, pending, mb, mr)
        assert 1 <= got <= min(len(pending), mb)


def test_padded_rows_take_the_index_stride():
    """hip.padded_rows(..., row_bytes=GpuIndex.row_bytes()): device-resident queries of an index whose bit rows sit at a 128-byte
    stride (768 bits = 96 bytes of data) are zero padded to it; rows that already fill their stride are left alone."""
    import numpy as np

    from lantern_amd import hip

    bits = np.arange(5 * 24, dtype=np.uint32).reshape(5, 24)
    plain = hip.padded_rows(bits, True)
    wide = hip.padded_rows(bits, True, row_bytes=128)
    assert plain.shape == (5, 24) and wide.shape == (5, 32) and wide.dtype == np.uint32
    assert np.array_equal(wide[:, :24], bits) and not wide[:, 24:].any()
    f = np.ones((3, 768), dtype=np.float32)
    assert hip.padded_rows(f, False, row_bytes=3072).shape == (3, 768)
    x = np.ones((2, 768), dtype=np.float32)
    x[0, ::2] = -1.0
    b = hip.padded_rows(x, False, b1=True, row_bytes=128)
    assert b.shape == (2, 128) and b.dtype == np.uint8 and b[0, 0] == 0b01010101 and b[1, 95] == 255 and not b[:, 96:].any()


def test_a_foreign_pointer_is_not_an_index_handle(capi):
    """Every entry point checks the handle's first word before dereferencing it as an index: a stale or foreign pointer gets an
    error string, not a walk through garbage (a mutex locked out of a string's bytes hangs the caller)."""
    junk = C.create_string_buffer(8192)
    err = C.c_char_p()
    assert capi.lib().usearch_size(C.cast(junk, C.c_void_p), C.byref(err)) == 0
    assert err.value and b"not an index handle" in err.value
