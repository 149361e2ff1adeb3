#!/bin/bash
# The kernel sources (ray_amd/csrc/rt_*.h) compiled for the host with AddressSanitizer + UndefinedBehaviorSanitizer, run
# through the whole host-build parity suite (counterpart of the reference's Asan / Tsan build configurations,
# CMakeLists.txt:16,30-49).  Out-of-bounds reads, misaligned or uninitialised accesses in a kernel body show up here, on
# the CPU, before they are silent on the device.  Restores the normal test library afterwards.
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd $REPO/tests/hostsim
FLAGS=$(python3 -c "import sys; sys.path.insert(0,'$REPO'); import __graft_entry__ as g; print(' '.join(f for f in g.HOSTSIM_FLAGS if not f.startswith('-O')))")
cp _build/libhostsim.so /tmp/libhostsim_plain.so
g++ $FLAGS -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer hostsim.cpp -o _build/libhostsim.so
cd $REPO
rm -f /tmp/asan_log* /tmp/ubsan_log*
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) \
ASAN_OPTIONS=detect_leaks=0:log_path=/tmp/asan_log UBSAN_OPTIONS=print_stacktrace=1:log_path=/tmp/ubsan_log \
    python -m pytest tests/test_hostsim_parity.py -x -q || true
cp /tmp/libhostsim_plain.so tests/hostsim/_build/libhostsim.so
if ls /tmp/asan_log* /tmp/ubsan_log* >/dev/null 2>&1; then
    echo "sanitizer reports:"; cat /tmp/asan_log* /tmp/ubsan_log* | head -100; exit 1
fi
echo "no sanitizer reports"
