#!/bin/bash
# Builds the library's host code with AddressSanitizer (device code untouched) together with
# tests/native/host_lifecycle.cpp and runs it.  CPU only.
set -e
cd "$(dirname "$0")/.."
OUT=/tmp/rebxamd_asan
mkdir -p $OUT
SRCS="reboundx_amd/csrc/stream_kernels.hip reboundx_amd/csrc/grfull_kernels.hip reboundx_amd/csrc/scan_kernels.hip reboundx_amd/csrc/seq_kernels.hip reboundx_amd/csrc/step_kernels.hip reboundx_amd/csrc/potential_kernels.hip reboundx_amd/csrc/operator_kernels.hip reboundx_amd/csrc/device_api.hip reboundx_amd/csrc/host_api.hip reboundx_amd/csrc/binary_io.hip"
/opt/rocm/bin/hipcc -O1 -g -std=c++17 --offload-arch=gfx950 -Xarch_host -fsanitize=address -Xarch_host -fno-omit-frame-pointer -Iinclude \
    $SRCS -x hip tests/native/host_lifecycle.cpp -o $OUT/host_lifecycle
ASAN_OPTIONS=detect_leaks=1 $OUT/host_lifecycle $OUT/state.rebx
