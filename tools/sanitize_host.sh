#!/bin/bash
# Host-side code under AddressSanitizer + UndefinedBehaviorSanitizer on the CPU (GPU ASan is not available on this
# pool): the C-ABI library's host half -- driver.cpp (CE / Nelder-Mead / PETS bookkeeping, table packing, argument checks) and multi.cpp,
# built with g++ -fsanitize=address,undefined and linked with the regular device objects -- and the C oracle, both exercised by the CPU
# test-suite (no GPU needed: the device-free entry points, the loader, the export list, the oracle's known-answer tests).
#   tools/sanitize_host.sh            -> builds into build/sanitize/, runs pytest, prints the summary; exit code = pytest's
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/build/sanitize
mkdir -p "$O"
make -C "$R/ratilqr.jl_amd/csrc" -s
CS=$R/ratilqr.jl_amd/csrc
FLAGS="-std=c++17 -O1 -g -fPIC -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include"
g++ $FLAGS -c "$CS/driver.cpp" -o "$O/driver_asan.o"
g++ $FLAGS -c "$CS/multi.cpp" -o "$O/multi_asan.o"
g++ -shared -fsanitize=address,undefined -o "$O/libratilqr_hip_asan.so" "$CS"/kernels_{sweep,roll,fused,b2,b3,b5,b8,misc,psw,bpsw,bpsw1}.o "$CS/sweep_dual.o" "$CS/wide.o" "$CS/wide32.o" \
    "$CS/ce_device.o" "$O/driver_asan.o" "$O/multi_asan.o" \
    -L/opt/rocm/lib -lamdhip64 -ldl -Wl,-rpath,/opt/rocm/lib
gcc -O1 -g -fPIC -std=c11 -fsanitize=address,undefined -fno-sanitize-recover=undefined -fopenmp -fno-fast-math -ffp-contract=off -shared \
    -o "$O/libratilqr_oracle_asan.so" "$R/oracle/ratilqr_oracle.c" -lm
cd "$R"
LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)" \
ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
RATILQR_SO="$O/libratilqr_hip_asan.so" RATILQR_ORACLE_SO="$O/libratilqr_oracle_asan.so" \
python -m pytest tests/test_cpu_abi.py tests/test_oracle_ileqg.py tests/test_oracle_ce.py tests/test_oracle_nm.py tests/test_oracle_pets.py \
    tests/test_oracle_leqg_identity.py tests/test_cpu_distributed.py -q -x "$@"
