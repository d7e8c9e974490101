"""Build recipe for the engine library (hipcc, gfx950 only) and the oracle checker."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libppcsr_hip.so")
ROOT = os.path.dirname(HERE)

HIPCC_FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared",
               # the rebalance position chain must round like the reference's x86-64 build: no FMA contraction
               "-ffp-contract=off",
               # (compact_block leaves its unrolled loops early on a wave-uniform bound: "loop not unrolled" is intended)
               "-Wno-pass-failed"]


def hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the engine is HIP-only (no CPU fallback)")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in os.listdir(CSRC) if f.endswith((".h", ".cc", ".hip")))


def build_engine(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    cmd = [hipcc()] + HIPCC_FLAGS + [os.path.join(CSRC, "ppcsr_hip.hip"), "-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=CSRC)
    return LIB


HOST_DIR = os.path.join(HERE, "host")
CLI = os.path.join(HOST_DIR, "ppcsr_cli")
CPP_TEST = os.path.join(ROOT, "tests", "cpp", "test_datastructure")
CPP_PAR_TEST = os.path.join(ROOT, "tests", "cpp", "test_parallel")


def build_host():
    """the C++ host side above the C ABI: CLI with the reference's flags + the restated DataStructureTest binary"""
    inc = ["-I" + os.path.join(ROOT, "include"), "-I" + HOST_DIR]
    link = ["-L" + CSRC, "-lppcsr_hip", "-Wl,-rpath," + CSRC]
    base = ["g++", "-std=c++17", "-O2", "-Wall", "-pthread"]
    subprocess.run(base + inc + [os.path.join(HOST_DIR, "main.cpp")] + link + ["-o", CLI], check=True)
    subprocess.run(base + inc + [os.path.join(ROOT, "tests", "cpp", "test_datastructure.cpp")] + link + ["-o", CPP_TEST], check=True)
    subprocess.run(base + inc + [os.path.join(ROOT, "tests", "cpp", "test_parallel.cpp")] + link + ["-o", CPP_PAR_TEST], check=True)
    return CLI


def build_oracle():
    """test infrastructure: the C restatement and (when /root/reference exists) the real reference"""
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s"], check=True)


if __name__ == "__main__":
    print(build_engine(force=True, verbose=True))
