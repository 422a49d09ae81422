#!/bin/bash
# The CPU suite (-m "not gpu") under a sanitizer build of the HOST side of the library, of the CLI and of the oracle (SURVEY 5;
# CPU only - GPU sanitizers are not available on this pool).
#   tools/run_sanitized.sh address,undefined [pytest args]
#   tools/run_sanitized.sh thread [pytest args]
# Everything instrumented shares ONE sanitizer runtime (ROCm clang's, -shared-libsan), preloaded into python so that the library
# it dlopens finds it initialised; leak checking is off (python itself never frees everything).
set -e -o pipefail
SAN=${1:-address,undefined}; shift || true
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TAG=${SAN//,/_}
CLANG=/opt/rocm/lib/llvm/bin/clang++
RT_DIR=$(dirname "$($CLANG -print-file-name=libclang_rt.asan-x86_64.so)")
case "$SAN" in
  thread) PRE=$RT_DIR/libclang_rt.tsan-x86_64.so ;;
  *) PRE=$RT_DIR/libclang_rt.asan-x86_64.so ;;
esac
make -C "$ROOT/microphaser_amd/csrc" -j8 SAN=$SAN > /tmp/mp_san_build_$TAG.log 2>&1 || { tail -30 /tmp/mp_san_build_$TAG.log; exit 1; }
make -C "$ROOT/oracle" SAN=$SAN CXX="$CLANG -shared-libsan" >> /tmp/mp_san_build_$TAG.log 2>&1 || { tail -30 /tmp/mp_san_build_$TAG.log; exit 1; }
export MP_LIB_DIR=$ROOT/microphaser_amd/_lib_san_$TAG
export MP_ORACLE_CLI=$ROOT/oracle/_build_san_$TAG/oracle_cli
export LD_LIBRARY_PATH=$RT_DIR:$LD_LIBRARY_PATH
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1:detect_odr_violation=0
export UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
export TSAN_OPTIONS="halt_on_error=0:second_deadlock_stack=1:exitcode=66:suppressions=$ROOT/tools/tsan.supp"
cd "$ROOT"
LD_PRELOAD=$PRE python -m pytest tests -x -q -m "not gpu" -p no:cacheprovider "$@"
