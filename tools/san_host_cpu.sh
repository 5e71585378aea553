#!/bin/bash
# AddressSanitizer + UBSan on the host-only C++ of the library (GFA/GAF ingest, dump, synthetic
# generator incl. text writer, malformed input): CPU only, run from the repository root.
# (An instrumented HOST build of the whole library was tried on the GPU box: the HIP runtime
#  itself does not come up under the preloaded sanitizer runtime on this pool.)
set -e
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -pthread -Iinclude -Iahsoka_amd/csrc \
    -o /tmp/ahsoka_san_test tools/san/san_main.cpp ahsoka_amd/csrc/ingest.cpp ahsoka_amd/csrc/synth.cpp
ASAN_OPTIONS=detect_leaks=1 UBSAN_OPTIONS=halt_on_error=1 /tmp/ahsoka_san_test
