#!/bin/bash
# Development aid: build an experimental librecur_amd.so with extra -D flags: mkabl.sh <name> <flags...>
# (the kernel files are recompiled with the flags, the host objects are build/obj's)
n=$1; shift
mkdir -p build/dev/abl_$n
for f in recur_amd/csrc/kernels_*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -Wno-unused-result -Wno-pass-failed "$@" -Irecur_amd/csrc -Iinclude -I/opt/rocm/include -c "$f" -o build/dev/abl_$n/"$(basename "$f" .hip)".o &
done
wait
host=$(make -s -C recur_amd/csrc print-host-objs)  # (not `ls build/obj`: an incremental tree keeps objects of files that no longer exist)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/dev/abl_$n/librecur_amd.so $host build/dev/abl_$n/kernels_*.o -Wl,-rpath,/opt/rocm/lib -lm -ldl
