"""The C ABI from a host with no Python in it: tests/capi/ip_search_host.cpp (HIP runtime + include/convdr_hip.h only) is
compiled with hipcc against the in-tree libconvdr_hip.so and run; it checks the exact top-k against its own fp64 brute
force.  This is the drop-in boundary as a C / C++ / cgo / JNI caller would use it."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_host_calls_the_c_abi_without_torch(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    lib_dir = os.path.join(ROOT, "convdr_amd")
    assert os.path.exists(os.path.join(lib_dir, "libconvdr_hip.so")), "build the library first (__graft_entry__.build())"
    exe = str(tmp_path / "ip_search_host")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "capi", "ip_search_host.cpp"),
                           "-o", exe, "-L" + lib_dir, "-lconvdr_hip", "-Wl,-rpath," + lib_dir])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "capi host ok (bf16 scan)" in out.stdout and "capi host ok (fp16 scan)" in out.stdout, out.stdout
