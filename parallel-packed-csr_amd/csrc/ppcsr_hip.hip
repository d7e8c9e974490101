// Single translation unit of libppcsr_hip.so (gfx950).  Build: see parallel-packed-csr_amd/build.py
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared ppcsr_hip.hip -o libppcsr_hip.so
// stable device radix sort (library op) for the PageRank consumer's transposition
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include "engine.cc"
#include "capi.cc"
