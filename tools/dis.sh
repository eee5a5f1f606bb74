#!/bin/bash
# Development aid: device ISA of the kernel files into build/dis/kernels_<part>.s (extra -D flags may follow)
cd "$(dirname "$0")/.." && mkdir -p build/dis
for f in recur_amd/csrc/kernels_*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -Wno-unused-result -Wno-pass-failed "$@" -Irecur_amd/csrc -Iinclude -I/opt/rocm/include --cuda-device-only -S "$f" -o build/dis/"$(basename "$f" .hip)".s 2>&1 | grep -v "argument unused" &
done
wait
