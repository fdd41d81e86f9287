#!/bin/bash
# The CPU side under AddressSanitizer + UndefinedBehaviorSanitizer: builds the sanitized host libraries and the
# oracle (`make sanitize`), runs host_selftest and the CPU test files that exercise them with the sanitizer runtime preloaded into
# python (MVS_BUILD_VARIANT=san makes the loaders take san/lib, san/bin and oracle/_build/san), then fuzzes the file readers for
# FUZZ_SECONDS (default 60).  Any sanitizer report fails the run.  usage: tools/run_sanitized.sh [pytest args]
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
make -C mesh-reconstruction_amd -j4 all >/dev/null
make -C mesh-reconstruction_amd -j4 sanitize
RT=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:detect_odr_violation=0
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export MVS_BUILD_VARIANT=san
mesh-reconstruction_amd/san/bin/host_selftest cpu tests/data/tracks | tail -2
LD_PRELOAD="$RT" python3 -m pytest -x -q -m "not gpu" tests/test_meshing_cpu.py tests/test_host_cpu.py tests/test_criteria_cpu.py tests/test_oracle_cpu.py tests/test_raster_cpu.py \
    tests/test_photometric_cpu.py tests/test_pinning_cpu.py "$@"
CORPUS=$(mktemp -d /tmp/mvs_fuzz_corpus_XXXXXX)
python3 tools/fuzz/make_seeds.py "$CORPUS" >/dev/null
(cd "$CORPUS" && "$ROOT/mesh-reconstruction_amd/san/bin/fuzz_readers" . -max_total_time="${FUZZ_SECONDS:-60}" -max_len=8192 -rss_limit_mb=3072 -malloc_limit_mb=1024 -timeout=10 \
    -print_final_stats=1 > fuzz.log 2>&1 || true; grep -E "^stat::number_of_executed|ERROR|SUMMARY|runtime error|^Done " fuzz.log || true
    grep -q "^Done " fuzz.log || { echo "the fuzzer did not run to its time limit:"; tail -5 fuzz.log; exit 1; }
    if ls crash-* leak-* oom-* timeout-* >/dev/null 2>&1; then echo "fuzzer findings:"; ls crash-* leak-* oom-* timeout-* 2>/dev/null; exit 1; fi)
rm -rf "$CORPUS"
echo "sanitized run: clean"
