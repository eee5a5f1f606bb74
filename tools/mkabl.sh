#!/bin/bash
# Development aid: build an experimental librecur_amd.so with extra -D flags: mkabl.sh <name> <flags...>
n=$1; shift
mkdir -p build/dev/abl_$n && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -Wno-unused-result -Wno-pass-failed "$@" -Irecur_amd/csrc -Iinclude -I/opt/rocm/include -c recur_amd/csrc/kernels.hip -o build/dev/abl_$n/kernels.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/dev/abl_$n/librecur_amd.so build/obj/rnn_core.o build/obj/dist.o build/obj/rnn_init.o build/obj/rnn_io.o build/obj/rnn_dump.o build/obj/cdb.o build/obj/charmodel.o build/obj/char_sampling.o build/obj/char_epoch.o build/obj/char_multitext.o build/obj/charmodel_meta.o build/obj/classify_host.o build/dev/abl_$n/kernels.o -Wl,-rpath,/opt/rocm/lib -lm -ldl
