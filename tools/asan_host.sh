#!/bin/bash
# asan_host.sh -- the host-side byte parsers and the CPU oracle under AddressSanitizer + UBSan, driven by tools/fuzz_host.cpp.
# CPU only (GPU ASan / XNACK are not available on this pool; none of this code touches a device): the container reader and the
# stream-format code are plain C++ headers (csrc/container.hpp, csrc/rc_format.hpp), the torchac-compatible coder and the file
# writer one plain C++ source (csrc/hostcoder.hip), the oracle plain C.
#   tools/asan_host.sh [fuzz_host arguments]        e.g.  tools/asan_host.sh --parse 1000000 --decode 100000 --coder 100000
# Build products go to tools/_asan/ (git-ignored).  Exit code 0 and a line "fuzz_host: ok ..." = zero reports.
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/tools/_asan"
mkdir -p "$OUT"
SAN="-fsanitize=address,undefined -fno-sanitize-recover=all -fno-omit-frame-pointer -g"
if [ "$OUT/fuzz_host" -ot "$ROOT/tools/fuzz_host.cpp" ] || [ "$OUT/fuzz_host" -ot "$ROOT/oracle/gpcc_oracle.c" ] || [ "$OUT/fuzz_host" -ot "$ROOT/gauspcc_amd/csrc/hostcoder.hip" ] \
   || [ "$OUT/fuzz_host" -ot "$ROOT/gauspcc_amd/csrc/container.hpp" ] || [ "$OUT/fuzz_host" -ot "$ROOT/gauspcc_amd/csrc/rc_format.hpp" ] || [ ! -x "$OUT/fuzz_host" ]; then
    gcc $SAN -O2 -std=gnu11 -fopenmp -ffp-contract=off -fno-fast-math -mavx2 -mfma -Wall -Wextra -Wno-unused-parameter -c "$ROOT/oracle/gpcc_oracle.c" -o "$OUT/gpcc_oracle.o"
    g++ $SAN -O1 -std=c++17 -Wall -x c++ -c "$ROOT/gauspcc_amd/csrc/hostcoder.hip" -o "$OUT/hostcoder.o"
    g++ $SAN -O1 -std=c++17 -Wall -c "$ROOT/tools/fuzz_host.cpp" -o "$OUT/fuzz_host.o"
    g++ $SAN -fopenmp "$OUT/fuzz_host.o" "$OUT/hostcoder.o" "$OUT/gpcc_oracle.o" -o "$OUT/fuzz_host" -lm -lpthread
fi
# leaks: the oracle's error returns drop what they had allocated (a test checker that stops at the first bad byte); memory SAFETY is what is asserted here
ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:allocator_may_return_null=1:max_allocation_size_mb=4096" UBSAN_OPTIONS="print_stacktrace=1" OMP_NUM_THREADS=1 "$OUT/fuzz_host" "$@"
