"""The GPU suite once more in the SHIPPED state of the library (round 6; VERDICT r5: "all 418 tests run the library in
its opt-in state").

tests/conftest.py sets HRFD_DEBUG_HOOKS=1 for the suite because many tests force failure and fallback paths through the
behaviour-changing hooks of include/hrfd_debug.h.  libhrfd reads that variable ONCE per process, so the shipped state
needs a fresh process: a child `python -m pytest tests -m gpu` with HRFD_HOOKS_OFF=1 (started as a subprocess of this
one -- never an exec of a process that has touched the GPU), in which every hook answers HRFD_ESTATE, the tests that
need one skip at the hook (conftest's autouse fixture), and everything else -- the reference-generated goldens, the
short-block sequences, BASELINE's shapes at full size with every channel an input of its own, the random walks, the
closed loop over the reference's count.raw, the ingest and fan-out transports, the shim classes -- runs on the dispatch a
user gets.  The child runs once per session; the tests below report it group by group (a group must have run a minimum
number of tests, none failed), and `test_shipped_state_summary` prints the child's own totals."""
import os
import subprocess
import sys
import xml.etree.ElementTree as ET

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (group, file, substring of the test name or "", least number of tests that must have PASSED without hooks)
GROUPS = [
    ("goldens_receive", "test_gpu_rx", "golden", 40),
    ("goldens_short_block_sequences", "test_gpu_short_blocks", "golden", 20),
    ("short_blocks_other", "test_gpu_short_blocks", "", 25),
    ("goldens_modulators", "test_gpu_tx_nco", "golden", 8),
    ("random_walks_receive", "test_gpu_rx", "random_walk", 12),
    ("random_walk_modulators", "test_gpu_tx_nco", "random_walk", 4),
    ("full_size_banks", "test_gpu_north_star_sizes", "", 8),
    ("full_size_bench_batch", "test_gpu_rx", "full_size", 1),
    ("fanout_and_config4", "test_fanout", "", 4),
    ("ingest_transport", "test_gpu_ingest", "", 3),
    ("count_raw_closed_loop", "test_count_raw", "gpu", 8),
    ("tools_playback_generators", "test_gpu_tools", "", 5),
    ("shim_classes", "test_shim", "", 5),
    ("modulators_nco", "test_gpu_tx_nco", "", 25),
    ("receive_everything_else", "test_gpu_rx", "", 120),
]


@pytest.fixture(scope="session")
def shipped_run(tmp_path_factory):
    """the child run: returns {(file, test name): outcome} and the child's last lines"""
    try:
        import torch
        torch.cuda.empty_cache()                          # this process has run most of the suite: hand its cached device memory back
    except Exception:
        pass
    xml = str(tmp_path_factory.mktemp("hooks_off") / "junit.xml")
    env = {k: v for k, v in os.environ.items() if k != "HRFD_DEBUG_HOOKS"}
    env["HRFD_HOOKS_OFF"] = "1"
    env.setdefault("HRFD_WALK_SEEDS", "8")
    cmd = [sys.executable, "-m", "pytest", os.path.join(ROOT, "tests"), "-m", "gpu", "-q", "-p", "no:cacheprovider",
           "--deselect", "tests/test_gpu_hooks_off.py", "--junitxml", xml]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    out = {}
    if os.path.exists(xml):
        for tc in ET.parse(xml).getroot().iter("testcase"):
            kind = "passed"
            for ch in tc:
                if ch.tag in ("failure", "error"):
                    kind = "failed"
                elif ch.tag == "skipped" and kind != "failed":
                    kind = "skipped"
            out[(tc.get("classname", ""), tc.get("name", ""))] = kind
    return out, r.returncode, (r.stdout[-1500:] + r.stderr[-500:])


def test_shipped_state_summary(shipped_run):
    out, rc, tail = shipped_run
    n = {k: sum(1 for v in out.values() if v == k) for k in ("passed", "skipped", "failed")}
    print(f"\nGPU suite without HRFD_DEBUG_HOOKS (the shipped state): {n}\n{tail[-600:]}")
    assert out, "the child run left no report:\n" + tail
    failed = [k for k, v in out.items() if v == "failed"]
    assert not failed and rc == 0, (failed[:10], tail)
    assert n["passed"] >= 280, n                       # (round 6: 297 passed, 196 skipped at the hooks they need)


@pytest.mark.parametrize("group,file,sub,least", GROUPS, ids=[g[0] for g in GROUPS])
def test_shipped_state(shipped_run, group, file, sub, least):
    out, rc, tail = shipped_run
    mine = {k: v for k, v in out.items() if k[0].endswith(file) and sub in k[1]}
    failed = [k[1] for k, v in mine.items() if v == "failed"]
    passed = sum(1 for v in mine.values() if v == "passed")
    assert not failed, (group, failed[:10])
    assert passed >= least, (group, passed, least, sum(1 for v in mine.values() if v == "skipped"))
