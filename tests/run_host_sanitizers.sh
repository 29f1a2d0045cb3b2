#!/bin/bash
# The HOST side of the library (URDF reader, model helpers, discretiser, argument checks, error paths, the C ABI's bookkeeping) under AddressSanitizer +
# UBSan: every source compiled with host instrumentation (`-fsanitize=address,undefined -fno-gpu-sanitize`: the device code is NOT instrumented -- GPU ASan is
# not available on this pool), linked into build/asan/libidocp_hip_asan.so, and the whole `-m "not gpu"` suite run against it (IDOCP_HIP_LIB).  CPU only.
# Round 6: 114 passed, no report (the URDF reader after its two fixes: unbounded recursion on a truncated closing tag, non-finite numbers accepted).
set -e
cd "$(dirname "$0")/.."
mkdir -p build/asan
rm -f build/asan/*.o
F="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer -Iinclude -Iidocp_amd/csrc"
for s in idocp_amd/csrc/*.hip idocp_amd/csrc/*.cpp; do
  x=""; case $s in *.hip) x="-x hip";; esac
  ( /opt/rocm/bin/hipcc $F $x -c $s -o build/asan/$(basename $s).o 2> build/asan/$(basename $s).log || echo "FAILED $s" ) &
  if (( $(jobs -r | wc -l) >= 6 )); then wait -n; fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fsanitize=address,undefined -fno-gpu-sanitize -shared -fPIC -o build/asan/libidocp_hip_asan.so build/asan/*.o
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
IDOCP_HIP_LIB=$PWD/build/asan/libidocp_hip_asan.so LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
  python -m pytest tests -x -q -m "not gpu" -p no:cacheprovider
