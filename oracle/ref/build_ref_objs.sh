#!/bin/bash
# Compile every reference translation unit of the listed library directories that builds against oracle/boost_shim
# (sources stay under $REFERENCE; objects go to oracle/_ref/obj) and archive them, so the drivers link lazily against
# exactly the reference objects they need.  TUs that need Boost components the shim does not cover (filesystem,
# spirit, math distributions, chrono/timer, program_options parsers) simply drop out; they are off the hot path.
set -u
REFERENCE=${REFERENCE:-/root/reference}
L=$REFERENCE/src/c++/lib
HERE=$(cd "$(dirname "$0")/.." && pwd)
OUT=$HERE/_ref
INC="-I$L -I$HERE/ref/gen -I$HERE/boost_shim -I$OUT/redist/htslib-1.7-6-g6d2bfb7 -I$OUT/redist/rapidjson-1.1.0/include -I$HERE/ref"
REF_OPT=${REF_OPT:--O3 -fomit-frame-pointer}
mkdir -p $OUT/obj
comp() {
    f=$1; o=$OUT/obj/$(echo ${f%.cpp} | tr '/' '_').o
    if [ "$o" -nt "$L/$f" ]; then return 0; fi
    g++ -std=c++11 $REF_OPT -w -fPIC -ffp-contract=off $INC -c $L/$f -o $o 2> $o.err || { rm -f $o; echo "skip $f: $(grep -m1 error $o.err)" >> $OUT/skipped.txt; }
}
export -f comp; export L INC OUT REF_OPT
rm -f $OUT/skipped.txt
(cd $L; ls blt_util/*.cpp blt_common/*.cpp common/*.cpp htsapi/*.cpp starling_common/*.cpp strelka_common/*.cpp alignment/*.cpp \
    calibration/*.cpp options/*.cpp appstats/*.cpp assembly/*.cpp applications/strelka/*.cpp applications/starling/*.cpp 2>/dev/null) | xargs -P ${JOBS:-8} -I{} bash -c 'comp {}'
rm -f $OUT/libreftus.a
ar rcs $OUT/libreftus.a $OUT/obj/*.o
echo "archived $(ls $OUT/obj/*.o | wc -l) reference objects; skipped $(wc -l < $OUT/skipped.txt 2>/dev/null || echo 0)"
