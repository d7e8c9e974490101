"""CPU: the host shims' locking under ThreadSanitizer (the reference builds its own tests a second time with
-fsanitize=thread, CMakeLists.txt:73-77).  tests/cpp/test_parallel.cpp — the reference's OpenMP tests restated with
std::thread — is compiled with -fsanitize=thread and linked against the CPU-emulator build of the SAME C ABI
(tests/hostsim/libppcsr_sim.so, test infrastructure): what is under test is the header-only host side (PCSR.h /
PPPCSR.h: the pending-batch mutex, the engine mutex, the lock stand-ins), not the engine."""
import os
import subprocess

from helpers import ROOT
from test_sim_engine import SIM_DIR, SIM_SO, build_sim

SRC = os.path.join(ROOT, "tests", "cpp", "test_parallel.cpp")
BIN = os.path.join(ROOT, "tests", "cpp", "test_parallel_tsan")


def test_parallel_suite_under_tsan():
    build_sim()
    inc = ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "parallel-packed-csr_amd", "host")]
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-pthread"] + inc + [SRC, SIM_SO,
                    "-Wl,-rpath," + SIM_DIR, "-o", BIN], check=True)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66 report_signal_unsafe=0", PPCSR_TEST_SMALL_ROUNDS="1")
    r = subprocess.run([BIN, "4", "20"], capture_output=True, text=True, timeout=900, env=env)
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
    assert r.returncode == 0 and "ALL PASSED" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
