#!/bin/bash
# Kernel-logic tests against the address/UB-sanitizer build of the CPU harness: the kernels'
# LDS is a heap block of exactly the size the real launch passes, so any out-of-range LDS or
# global access in the kernel source is reported (GPU ASan is not available on the pool).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
make -C "$ROOT/subpixal_amd/csrc" emu-asan
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
cd "$ROOT"
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
LD_PRELOAD="$RT" SPX_EMU_LIB="$ROOT/tests/cpu_emu/libspx_emu_asan.so" \
python -m pytest tests/test_kernel_logic_cpu.py -x -q "$@"
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
LD_PRELOAD="$RT" SPX_EMU_LIB="$ROOT/tests/cpu_emu/libspx_emu_asan.so" python tools/emu_sanitizer_cases.py
