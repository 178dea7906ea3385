#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the HOST side of the library (CPU only: GPU ASan is
# not available on this pool): csrc/fastx.cpp (FASTA/FASTQ parser, reader threads) and csrc/capi.cpp
# (planning, file headers, argument checks) are rebuilt instrumented and linked with the kernels as built;
# then the CPU test cases that drive them run against that library.
#     tools/sanitize_host.sh            (from the repository root)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
python3 -m btl_bloomfilter_amd.build > /dev/null
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
SAN="-fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer -g -O1"
P=btl_bloomfilter_amd
mkdir -p $P/_build_asan
for u in capi fastx; do
	$HIPCC --offload-arch=gfx950 -std=c++17 -fPIC -Wall -Wno-unused-function $SAN -c -o $P/_build_asan/$u.o $P/csrc/$u.cpp
done
$HIPCC --offload-arch=gfx950 -shared -fPIC $SAN -o $P/libbtlbf_asan.so $P/_build_asan/capi.o $P/_build_asan/fastx.o \
	$P/_build/seq_kernels.o $P/_build/aux_kernels.o $P/_build/partition_kernels.o $P/_build/part_hash_h*.o -lz
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
[ -f "$RT" ] || RT=$(/opt/rocm/lib/llvm/bin/clang --print-file-name=libclang_rt.asan.so)
echo "runtime: $RT"
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
	BTLBF_LIB=$ROOT/$P/libbtlbf_asan.so python3 -m pytest tests/test_fastx.py tests/test_abi_cpu.py -q -m "not gpu" -p no:cacheprovider
