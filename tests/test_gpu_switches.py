"""Every diagnostic LANECV_* switch that changes what is launched must leave the results alone: the same batches through
a six-slot pipeline in a process of its own per setting (most switches are read once per process), each checked
against the oracle by tests/switch_child.py."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
SETTINGS = [
    {},
    {"LANECV_FRONT_WAVES": "2"}, {"LANECV_FRONT_WAVES": "4"},
    {"LANECV_FRONT_ROWS": "40"},
    {"LANECV_FRONT_TILED": "1"},
    {"LANECV_NO_ROW_SKIP": "1"}, {"LANECV_NO_STRIP_SHIFT": "1"},
    {"LANECV_PPHT_WAVES": "8"},
    {"LANECV_PPHT_LDS_KB": "49.5"}, {"LANECV_PPHT_LDS_KB": "16"},
    {"LANECV_PPHT_TWO_PER_CU": "0"}, {"LANECV_NO_SCATTER_KERNEL": "1"}, {"LANECV_PPHT_FULL_AT_ONCE": "1"},
    {"LANECV_VOTE_FUSED": "0"},
    {"LANECV_NO_GROUPS": "1", "LANECV_RESULT_COPY": "1"},
    {"LANECV_LINES_STREAMS": "1"}, {"LANECV_LINES_PRIORITY": "0"}, {"LANECV_POST_STREAM": "0", "LANECV_TAIL_STREAMS": "0"},
    {"LANECV_QUEUE_PROBE": "0"},
    {"LANECV_FULL_UPLOAD": "1"}, {"LANECV_NO_UPLOAD_WINDOWS": "1"},
    {"LANECV_EAGER_REDO": "1"}, {"LANECV_EAGER_REDO": "1", "LANECV_REDO_ON_EDGE": "1"},
]


def test_switches_leave_results_alone():
    failed = []
    for s in SETTINGS:
        env = dict(os.environ)
        env.update(s)
        env["LANECV_SKIP_OFF_BATCHES"] = "0"
        r = subprocess.run([sys.executable, os.path.join(HERE, "switch_child.py")], env=env, capture_output=True, text=True, timeout=110)
        if r.returncode != 0 or "OK" not in r.stdout:
            failed.append((s, r.returncode, (r.stderr or "")[-600:]))
    assert not failed, failed
