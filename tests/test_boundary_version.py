"""CPU suite, part 6: the boundary is versioned and the exchange's retry rule.  No GPU."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(ROOT, "codex-storage-proofs-circuits_amd")


def header_version():
    header = open(os.path.join(ROOT, "include", "codex_p2.h")).read()
    return (int(re.search(r"^#define CP2_ABI_VERSION_MAJOR (\d+)", header, re.M).group(1)),
            int(re.search(r"^#define CP2_ABI_VERSION_MINOR (\d+)", header, re.M).group(1)))


def test_library_python_nim_and_cpp_state_the_headers_version(pkg):
    """include/codex_p2.h carries CP2_ABI_VERSION_MAJOR / _MINOR; the library answers cp2_abi_version() without a device and its
    SONAME carries the major; every binding in the tree states the version it was written against and checks it before its first
    real call: ctypes, the Nim binding (checked as text: no Nim compiler here), the C++ mirror and with it the cli twin."""
    major, minor = header_version()
    L = pkg.load_library()
    assert L.cp2_abi_version() == (major << 16) | minor
    assert (pkg.ABI_VERSION_MAJOR, pkg.ABI_VERSION_MINOR) == (major, minor)
    dyn = subprocess.run(["readelf", "-d", pkg.LIB_PATH], capture_output=True, text=True).stdout
    assert "Library soname: [libcodex_p2.so.%d]" % major in dyn, dyn
    assert os.path.exists(pkg.LIB_PATH + ".%d" % major)                       # what a linked program looks for beside itself
    needed = subprocess.run(["readelf", "-d", pkg.CLI_PATH], capture_output=True, text=True).stdout
    assert "[libcodex_p2.so.%d]" % major in needed, needed
    nim = open(os.path.join(PKG_DIR, "nim", "codex_p2.nim")).read()
    assert int(re.search(r"abiVersionMajor\* = (\d+)", nim).group(1)) == major
    assert int(re.search(r"abiVersionMinor\* = (\d+)", nim).group(1)) == minor
    assert "proc cp2_abi_version(): cint {.importc.}" in nim and re.search(r"requireAbi\(\)\s*\n\s*let st = cp2_multi_init", nim)
    mirror = open(os.path.join(PKG_DIR, "host", "proof_input_api.hpp")).read()
    assert re.search(r"requireAbi\(\);\s*\n\s*int st = cp2_multi_init", mirror)


def test_a_library_of_another_major_is_refused_with_both_numbers(pkg, tmp_path):
    """ctypes: a stand-in library that exports nothing but cp2_abi_version (major + 1), named through the A/B override.  C++ mirror /
    cli twin: a program built against this header that finds, at run time, a libcodex_p2.so.<major> answering major + 1."""
    major, minor = header_version()
    src = tmp_path / "other.c"
    src.write_text("int cp2_abi_version(void) { return (%d << 16) | 0; }\n" % (major + 1))
    so = str(tmp_path / "libother.so")
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-o", so, str(src)])
    code = ("import sys; sys.path.insert(0, %r); import __graft_entry__ as g; p = g.load_package()\n"
            "try:\n    p.load_library()\nexcept RuntimeError as e:\n    print('refused:', e)" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CODEX_P2_LIB=so), capture_output=True, text=True, timeout=120)
    assert "refused:" in r.stdout and "%d.0" % (major + 1) in r.stdout and "%d.%d" % (major, minor) in r.stdout, (r.stdout, r.stderr[-500:])

    prog = tmp_path / "mirror.cpp"
    prog.write_text('#include <cstdio>\n#include "%s"\n'
                    'int main() { try { codex::Engine e; } catch (const std::exception& e) { std::printf("refused: %%s\\n", e.what()); return 3; } return 0; }\n'
                    % os.path.join(PKG_DIR, "host", "proof_input_api.hpp"))
    fake = tmp_path / "fake"
    fake.mkdir()
    stub = tmp_path / "stub.c"
    names = [n for n in pkg.exported_symbols() if n != "cp2_abi_version"]
    stub.write_text("int cp2_abi_version(void) { return (%d << 16) | 0; }\n" % (major + 1) + "".join("void %s(void) {}\n" % n for n in names))
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-Wl,-soname,libcodex_p2.so.%d" % major, "-o", str(fake / ("libcodex_p2.so.%d" % major)), str(stub)])
    exe = str(tmp_path / "mirror")
    libdir = os.path.dirname(pkg.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-o", exe, str(prog), "-L" + libdir, "-lcodex_p2", "-pthread"])
    r = subprocess.run([exe], env=dict(os.environ, LD_LIBRARY_PATH=str(fake)), capture_output=True, text=True, timeout=120)
    assert r.returncode == 3 and "ABI version %d.0" % (major + 1) in r.stdout and "built against %d.%d" % (major, minor) in r.stdout, (r.returncode, r.stdout, r.stderr)


POLICY_CPP = r"""
#include "exchange_policy.hpp"
#include <cstdio>
#include <initializer_list>
using cp2i::exchange_may_retry_on_host;
int main() {
  int bad = 0, yes = 0;
  for (int status : {CP2_OK, CP2_ERR_INVALID, CP2_ERR_HIP, CP2_ERR_ALLOC, CP2_ERR_IO})
    for (int timed_out = 0; timed_out < 2; ++timed_out)
      for (int mode : {CP2_GATHER_AUTO, CP2_GATHER_RCCL, CP2_GATHER_HOST, CP2_GATHER_COPY})
        for (size_t world : {(size_t)1, (size_t)2, (size_t)8})
          for (int attempt = 0; attempt < 2; ++attempt) {
            const bool want = status == CP2_ERR_HIP && !timed_out && mode == CP2_GATHER_AUTO && world > 1 && attempt == 0;
            const bool got = exchange_may_retry_on_host(status, timed_out != 0, mode, world, attempt);
            if (got != want) ++bad;
            yes += got;
          }
  std::printf("policy %s (%d combinations retried)\n", bad ? "WRONG" : "ok", yes);
  return bad != 0;
}
"""


def test_exchange_never_retries_through_host_memory_after_a_timeout(tmp_path):
    """csrc/exchange_policy.hpp, the header csrc/multi_gpu.cpp decides with: a launch error or a failed verification in the automatic
    mode is retried once through host memory; a TIME-OUT never is -- the collective is still queued on the very streams the host path
    would enqueue on and then wait for without a bound.  Compiled and walked over every combination on the CPU; the GPU suite runs a
    collective that really does not complete (tests/test_gpu_round6.py)."""
    src = tmp_path / "policy.cpp"
    src.write_text(POLICY_CPP)
    exe = str(tmp_path / "policy")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-I" + os.path.join(PKG_DIR, "csrc"), "-o", exe, str(src)])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "policy ok (2 combinations retried)" in r.stdout, (r.stdout, r.stderr)
