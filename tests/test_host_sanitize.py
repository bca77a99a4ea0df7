"""Host-only sanitizer builds of the C ABI's host code (SURVEY.md section 5 'race detection / sanitizers'): the batch
validation, the 8-thread block-diagonal CSR / transposed-CSR build and the row tiers of mllp_amd/csrc/host_graph.cpp,
compiled with g++ -fsanitize=address,undefined and -fsanitize=thread and driven by host_graph_test.cpp on ragged
random batches, plus the MPS reader (mps_reader.cpp) on the committed fixtures and a synthetic file.  No GPU, no HIP runtime."""
import os
import subprocess

import pytest

CSRC = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "mllp_amd", "csrc"))


@pytest.fixture(scope="module")
def built():
    r = subprocess.run(["make", "-C", CSRC, "host-sanitize"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("binary", ["host_graph_asan", "host_graph_tsan"])
def test_host_graph_build_is_clean_under_sanitizers(built, binary):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", TSAN_OPTIONS="halt_on_error=1")
    mps = sorted(os.path.join(CSRC, "..", "..", "tests", "golden", "mps", f)
                 for f in os.listdir(os.path.join(CSRC, "..", "..", "tests", "golden", "mps")))
    r = subprocess.run([os.path.join(CSRC, binary)] + mps, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-6000:]
    assert "ok" in r.stdout and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr
