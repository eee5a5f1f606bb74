#!/bin/bash
# Development aid: device ISA of kernels.hip into build/dis/kernels.s (extra -D flags may follow)
cd "$(dirname "$0")/.." && mkdir -p build/dis && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -Wno-unused-result -Wno-pass-failed "$@" -Irecur_amd/csrc -Iinclude -I/opt/rocm/include --cuda-device-only -S recur_amd/csrc/kernels.hip -o build/dis/kernels.s 2>&1 | grep -v "argument unused"
