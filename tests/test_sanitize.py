"""Host-only AddressSanitizer + UndefinedBehaviorSanitizer run of the COCO RLE codec (VERDICT r01 robustness: a counts
string with 13+ continuation groups shifted past the word).  rle_host.hip is plain C++, so it is built with g++ -fsanitize and
driven by tests/sanitize/rle_sanitize_main.cpp: round trips on random masks / polygons and hostile inputs.  GPU sanitizers are not
available on the pool; device code is covered by the parity tests instead."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rle_codec_is_clean_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / "rle_sanitize")
    rocm_inc = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "include")
    cmd = ["g++", "-x", "c++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-D__HIP_PLATFORM_AMD__", "-I" + rocm_inc, "-o", exe,
           os.path.join(ROOT, "tests", "sanitize", "rle_sanitize_main.cpp"), os.path.join(ROOT, "ampis_amd", "csrc", "rle_host.hip")]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "RLE SANITIZE OK" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
