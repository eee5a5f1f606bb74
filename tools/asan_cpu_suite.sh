#!/bin/bash
# Build container: the library's host code (gnu11 C) under AddressSanitizer + UBSan, the CPU test suite against it.
# (GPU AddressSanitizer is not available on the pool; the kernel objects are the ordinary ones.)
#   tools/asan_cpu_suite.sh        -> prints the sanitizers' reports, if any, and the suite's verdict
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
cd "$root" && make -C recur_amd/csrc > /dev/null
mkdir -p build/asan
for f in $(make -s -C recur_amd/csrc print-c-srcs | sed "s/\.c//g"); do
  gcc -std=gnu11 -O1 -g -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -D_GNU_SOURCE -Irecur_amd/csrc -Iinclude \
      -I/opt/rocm/include -c recur_amd/csrc/$f.c -o build/asan/$f.o
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -o build/asan/librecur_amd.so $(for f in $(make -s -C recur_amd/csrc print-c-srcs | sed "s/\.c//g"); do echo build/asan/$f.o; done) \
    build/obj/kernels_forward.o build/obj/kernels_loss.o build/obj/kernels_chain.o build/obj/kernels_bptt.o build/obj/kernels_apply.o build/obj/kernels_support.o -Wl,-rpath,/opt/rocm/lib -lm -ldl
export RECUR_AMD_LIB=$root/build/asan/librecur_amd.so
export LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=0 UBSAN_OPTIONS=print_stacktrace=1
python -m pytest tests/ -q -m "not gpu" -p no:cacheprovider -s 2>&1 | grep -i "AddressSanitizer\|runtime error\|SUMMARY\| passed\| failed" | sort | uniq -c
