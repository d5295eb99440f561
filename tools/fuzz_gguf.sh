#!/bin/bash
# Builds tools/fuzz_gguf.cpp + the host GGUF reader with ASan/UBSan (CPU only) and runs it over the
# committed fixtures.  usage: tools/fuzz_gguf.sh [iterations]
set -euo pipefail
cd "$(dirname "$0")/.."
P=bitnet-rs_amd
OUT=/tmp/fuzz_gguf_$$
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -Wno-unused-result \
    -D__HIP_PLATFORM_AMD__ -Iinclude -I$P/host -I/opt/rocm/include \
    tools/fuzz_gguf.cpp $P/host/gguf.cpp $P/host/decoder.cpp -o $OUT \
    -L$P -lbitnet_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/$P -Wl,-rpath,/opt/rocm/lib
ASAN_OPTIONS=detect_leaks=1:allocator_may_return_null=1 $OUT "${1:-200000}" tests/golden/*.gguf
rm -f $OUT
