"""Host code under AddressSanitizer + UndefinedBehaviorSanitizer (no GPU needed): the C-ABI's argument
validation in pt_capi.hip (host side compiled with the sanitizers, device code as always), the C++ scene
API, the .dat/.raw loader and the oracle, driven by tests/cpp/sanitize_driver.cpp; plus the pytest probe."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:protect_shadow_gap=0", UBSAN_OPTIONS="print_stacktrace=1")


def test_host_side_is_clean_under_asan_and_ubsan():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp"), "asan"], check=True, timeout=600)
    out = subprocess.run([os.path.join(ROOT, "tests", "cpp", "_asan", "sanitize_driver")], env=ENV, capture_output=True,
                         text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "sanitize ok" in out.stdout
    assert "ERROR: AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr, out.stderr
    probe = subprocess.run([os.path.join(ROOT, "tests", "cpp", "_asan", "host_probe")], input="layout\naxisangle 2 1 0 0.5\n",
                           env=ENV, capture_output=True, text=True, timeout=60)
    assert probe.returncode == 0 and probe.stdout.startswith("32 64 80 32 64 144"), probe.stdout + probe.stderr
    assert "runtime error" not in probe.stderr


def test_oracle_through_ctypes_is_clean_under_asan_and_ubsan():
    """The checker matters as much as the product: what the GPU suite's first tests do with the oracle in the test
    process (tests/replay_oracle_ctypes.py: the demo scene's orbit frames with and without a short volume, row ranges,
    no float frame, samples, trilinear, every probe) with the oracle built under ASan + UBSan and gcc's sanitizer
    runtimes preloaded into Python."""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp"), "oracle-asan"], check=True, timeout=600)
    libs = [subprocess.run(["gcc", f"-print-file-name={n}"], capture_output=True, text=True, check=True).stdout.strip()
            for n in ("libasan.so", "libubsan.so")]
    assert all(os.path.isabs(p) and os.path.exists(p) for p in libs), libs
    env = dict(ENV, UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", LD_PRELOAD=":".join(libs),
               PT_ORACLE_LIB=os.path.join(ROOT, "tests", "cpp", "_asan", "libpt_oracle_asan.so"))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "replay_oracle_ctypes.py")], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "replay ok" in out.stdout, out.stdout + out.stderr
    assert "ERROR: AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr, out.stderr


def test_the_library_in_the_tree_was_built_from_these_sources():
    import __graft_entry__ as g

    assert os.path.exists(g.LIB), "libpt_hip.so is missing: python __graft_entry__.py"
    assert g.built_hash() == g.source_hash(), "libpt_hip.so is stale: rebuild with python __graft_entry__.py"
