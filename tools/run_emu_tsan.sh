#!/bin/bash
# Race detection for the kernels' synchronisation: the CPU harness runs every work-item as a
# thread, block_sync/wave_sync as std::barrier; under ThreadSanitizer a missing barrier in the
# kernel source (e.g. between the plane writes and the coarse arg-max, or across the pair
# boundary that has no end-of-pair barrier) is reported as a data race.  Exit code 1 on any report.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
make -C "$ROOT/subpixal_amd/csrc" emu-tsan
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.tsan-x86_64.so | head -1)
cd "$ROOT"
LOG=$(mktemp)
TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 history_size=4" LD_PRELOAD="$RT" \
SPX_EMU_LIB="$ROOT/tests/cpu_emu/libspx_emu_tsan.so" python tools/emu_sanitizer_cases.py --quick > "$LOG" 2>&1 || { tail -30 "$LOG"; exit 1; }
grep -v "^==\|^$" "$LOG" | tail -12
N=$(grep -c "WARNING: ThreadSanitizer" "$LOG" || true)
echo "ThreadSanitizer reports: $N"
test "$N" = "0"
