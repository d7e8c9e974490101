// TEST INFRASTRUCTURE ONLY: the engine's kernel + host source compiled for the CPU SIMT emulator
// (sim_runtime.*).  Produces tests/hostsim/libppcsr_sim.so which only tests/test_sim_*.py load, to debug
// kernel logic without a GPU.  The product library is built from csrc/ppcsr_hip.hip by hipcc and contains
// none of this.
#define PPCSR_SIM 1
#include "engine.cc"
#include "capi.cc"
