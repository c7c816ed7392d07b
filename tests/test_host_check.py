"""CPU suite, part 4: the DEVICE arithmetic source (csrc/fr_gfx950.hpp, poseidon2_dev.hpp) compiled for the
host with CP2_HOST_CHECK: 128-bit shadow accumulator (traps on 64-bit column overflow), asserted limb/value
bounds of the lazy reduction, AddressSanitizer + UBSan, every result compared with the C oracle."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_device_arithmetic_on_host_with_sanitizers(tmp_path):
    exe = str(tmp_path / "host_check")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-o", exe, os.path.join(ROOT, "tests", "host_check", "host_check.cpp"),
                           os.path.join(ROOT, "oracle", "p2_oracle.c"), "-lpthread"])
    r = subprocess.run([exe, "20000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert ", 0 mismatches, no bound violations" in r.stdout
