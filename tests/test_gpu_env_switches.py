"""Every result-neutral environment switch of the library, once, against the default configuration: the same graph
checksums and the same answers, bit for bit, from tests/env_switch_probe.py run in a process of its own per switch (several switches
are read once per process).  The tested configuration space is the default plus each switch alone -- what a deployment can reach by
setting one variable."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# switch(es) -> value.  Not here: deployment switches that do not change a kernel (DEVICE, RCCL_LIB, SAVE_CHUNK_BYTES, SCAN_*: covered by
# tests/test_scan_server.py and tests/test_gpu_fake_rccl.py), diagnostics, and LANTERN_GPU_PQ_ADC (a different summation order by
# definition: its own parity test, test_compact_pq_index_searches_by_adc_over_the_code_bytes).
SWITCHES = [{"LANTERN_GPU_SPEC": "0"}, {"LANTERN_GPU_SPEC": "1"}, {"LANTERN_GPU_SPEC": "2"}, {"LANTERN_GPU_SPEC": "3"}, {"LANTERN_GPU_SPEC": "4"},
            {"LANTERN_GPU_SOLO": "1"}, {"LANTERN_GPU_SPEC_WAVES": "6"}, {"LANTERN_GPU_LDS_LIST": "1"}, {"LANTERN_GPU_WIDE_ROWS": "0"}, {"LANTERN_GPU_WIDE_ROWS": "1"},
            {"LANTERN_GPU_VIS_SLOTS": "0"}, {"LANTERN_GPU_VIS_SLOTS": "256"}, {"LANTERN_GPU_INSERT_VIS_SLOTS": "0"}, {"LANTERN_GPU_INSERT_VIS_SLOTS": "256"},
            {"LANTERN_GPU_WAVES_PER_CU": "8"},
            {"LANTERN_GPU_TICKETS": "0"}, {"LANTERN_GPU_INSERT_SPEC": "0"}, {"LANTERN_GPU_REPRUNE_STATE": "0"}, {"LANTERN_GPU_REGS_CPL4": "1"},
            {"LANTERN_GPU_GROUP_ALL": "1"}, {"LANTERN_GPU_DENSE_FUSED": "0"}, {"LANTERN_GPU_ADC_SPEC": "0"}, {"LANTERN_GPU_PQ_COMPACT": "1"},
            {"LANTERN_GPU_GATHER_WALKSHAPE": "1"}, {"LANTERN_GPU_NOTIFY_SPIN_US": "0"}, {"GPU_MAX_HW_QUEUES": "4"},
            # [r6] the visited bitmap's undo log (walk.hpp VisUndo): a log of 16 entries overflows in every walk that reaches the bitmap, so these
            # run the clear-at-the-end path -- after a spill, in bitmap-only mode, in the search walks and in the insertion walks, in both the
            # bandwidth-bound and the latency-bound shapes; a bitmap left dirty by one walk would change the next walk of its workgroup
            {"LANTERN_GPU_VIS_SLOTS": "256", "LANTERN_GPU_VIS_UNDO": "16"}, {"LANTERN_GPU_VIS_SLOTS": "0", "LANTERN_GPU_VIS_UNDO": "16"},
            {"LANTERN_GPU_VIS_SLOTS": "256", "LANTERN_GPU_VIS_UNDO": "0"},
            {"LANTERN_GPU_INSERT_VIS_SLOTS": "256", "LANTERN_GPU_VIS_UNDO": "16"}, {"LANTERN_GPU_INSERT_VIS_SLOTS": "0", "LANTERN_GPU_VIS_UNDO": "16"},
            {"LANTERN_GPU_VIS_SLOTS": "256", "LANTERN_GPU_SPEC": "0"}, {"LANTERN_GPU_VIS_SLOTS": "256", "LANTERN_GPU_SPEC": "1", "LANTERN_GPU_VIS_UNDO": "16"},
            {"LANTERN_GPU_VIS_SLOTS": "0", "LANTERN_GPU_SPEC": "3"}, {"LANTERN_GPU_VIS_SLOTS": "256", "LANTERN_GPU_SPEC": "3", "LANTERN_GPU_VIS_UNDO": "16"}]


def probe(extra):
    env = {k: v for k, v in os.environ.items() if not k.startswith("LANTERN_GPU_")}
    env.update(extra)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "env_switch_probe.py")], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    line = next((json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")), None)
    assert p.returncode == 0 and line, (extra, p.stdout[-1500:], p.stderr[-1500:])
    return line


@pytest.fixture(scope="module")
def default_line():
    from lantern_amd import build, capi

    build.build()
    assert capi.device_count() > 0
    line = probe({})
    assert line["pq_compact_equals_expanded"], "a compact pq index that decodes rows on the fly must answer as the expanded one"
    return line


@pytest.mark.parametrize("switch", SWITCHES, ids=[",".join(f"{n}={v}" for n, v in sw.items()) for sw in SWITCHES])
def test_switch_is_result_neutral(default_line, switch):
    assert probe(switch) == default_line, f"{switch} changed a graph or an answer"
