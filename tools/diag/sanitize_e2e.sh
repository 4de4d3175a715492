#!/bin/bash
# The adapter, the hooked translation units and the library's host stages (strelka_amd/host/*.cpp through the CPU double of the C-ABI)
# under AddressSanitizer + UndefinedBehaviorSanitizer (halt on the first finding), over seeded end-to-end samples compared with the
# reference byte for byte (tools/fuzz/e2e_seeds.py).  Build container only (needs /root/reference); everything is built under
# /tmp/asan -- the shipped binaries under oracle/_ref/bin are not touched (two symbolic links *_asan exist there while this runs).
#   tools/diag/sanitize_e2e.sh [germline seeds=6] [two-sample seeds=3] [tumour / normal seeds=3] [hard seeds=4]
set -e
cd "$(dirname "$0")/../.."
ROOT=$PWD
SAN="-O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer"
mkdir -p /tmp/asan
make -s -C adapter double OBJ=/tmp/asan/obj BIN=/tmp/asan/bin REF_OPT="$SAN" -j16
gcc -std=c11 -fPIC -ffp-contract=off $SAN -c oracle/strelka_oracle.c -o /tmp/asan/oracle_for_double.o
g++ -std=c++17 -fPIC -ffp-contract=off $SAN -shared -Ioracle -Iinclude -Istrelka_amd/csrc oracle/abi_double.cpp strelka_amd/host/*.cpp \
    /tmp/asan/oracle_for_double.o -lm -lpthread -lz -o /tmp/asan/libstrelka_amd_double.so
ln -sf /tmp/asan/bin/starling2_dbl oracle/_ref/bin/starling2_asan
ln -sf /tmp/asan/bin/strelka2_dbl oracle/_ref/bin/strelka2_asan
trap 'rm -f "$ROOT"/oracle/_ref/bin/starling2_asan "$ROOT"/oracle/_ref/bin/strelka2_asan' EXIT
export LD_LIBRARY_PATH=/tmp/asan:$ROOT/oracle ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
python tools/fuzz/e2e_seeds.py ${1:-6} 121 asan 6 | tail -n 1
python tools/fuzz/e2e_seeds.py ${2:-3} 5 asan 3 multi | tail -n 1
python tools/fuzz/e2e_seeds.py ${3:-3} 9 asan 3 somatic | tail -n 1
SK_FUZZ_HARD=1 python tools/fuzz/e2e_seeds.py ${4:-4} 2011 asan 4 | tail -n 1
