"""bench.py's evidence plumbing, on the CPU: the device-code tag that ties a PMC summary to the code it measured, the
per-workload traffic lookup, the workload names shared by the `also` entries and profiles/latest_pmc_traffic.json."""
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_kernel_code_tag_is_the_hash_of_the_device_code(bench):
    """sha256 of libhrfd.so's .hip_fatbin section: 16 hex digits, stable from call to call, and NOT a function of
    the sources' comments (that was the round-3 tag)"""
    tag = bench.kernel_code_tag()
    assert len(tag) == 16 and int(tag, 16) >= 0
    assert tag == bench.kernel_code_tag()
    assert tag != bench.kernel_source_tag()


def test_workload_names(bench):
    assert bench.rx_workload_name("wbfm", 256, 16) == "wbfm_256x16"
    assert bench.rx_workload_name("wbfm", 256, 16, "random") == "wbfm_256x16_random"
    assert bench.rx_workload_name("wbfm", 256, 16, "fmtone", 0.25) == "wbfm_256x16_quiet25"
    assert bench.rx_workload_name("mixed", 256, 16, iqdump=True) == "mixed_256x16_iqdump"


def test_committed_pmc_summary_is_well_formed_and_reported_only_for_its_code(bench, monkeypatch):
    with open(os.path.join(ROOT, "profiles", "latest_pmc_traffic.json")) as f:
        summ = json.load(f)
    assert len(summ["kernel_code_tag"]) == 16
    for name in ("wbfm_256x16", "wbfm_1024x16", "mixed_256x16", "ssbmod_1024x16", "wbfm_256x16_iqdump", "wbfm_256x16_quiet25"):
        w = summ["workloads"][name]
        assert w["FETCH_SIZE_KiB"] > 0 and w["WRITE_SIZE_KiB"] > 0
    # the headline workload: FETCH x 2 + WRITE within 1 % of the algorithmic bytes (2.0078 B per IQ sample)
    w = summ["workloads"]["wbfm_256x16"]
    algo = 256 * 16 * (262144 + 1024 + 4)
    assert abs((w["FETCH_SIZE_KiB"] * 2048 + w["WRITE_SIZE_KiB"] * 1024) / algo - 1.0) < 0.01
    # config 5: what the SSB modulator writes is its output, to a part in a thousand
    w = summ["workloads"]["ssbmod_1024x16"]
    assert abs(w["WRITE_SIZE_KiB"] * 1024 / (1024 * 16 * 512 * 512) - 1.0) < 0.002
    # the lookup: the summary's own code -> bytes; any other code -> None with the reason
    bench._PMC = None
    monkeypatch.setattr(bench, "kernel_code_tag", lambda: summ["kernel_code_tag"])
    t, src = bench.pmc_traffic("wbfm_256x16")
    assert t == int(summ["workloads"]["wbfm_256x16"]["FETCH_SIZE_KiB"] * 2048 + summ["workloads"]["wbfm_256x16"]["WRITE_SIZE_KiB"] * 1024)
    assert bench.pmc_traffic("no_such_workload")[0] is None
    monkeypatch.setattr(bench, "kernel_code_tag", lambda: "0" * 16)
    t, src = bench.pmc_traffic("wbfm_256x16")
    assert t is None and "not reported" in src
    bench._PMC = None
