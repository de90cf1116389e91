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
# the image decoders on their own: progressive and baseline JPEG, interlaced / 16-bit / palette PNG (files of tests/golden/jpeg_cases.npz)
python3 - "$R" "$W" <<'EOP'
import sys, numpy as np
Z = np.load(sys.argv[1] + "/tests/golden/jpeg_cases.npz")
want = ["progressive_420_57x33", "progressive_gray_17x29", "progressive_422_40x21", "restart_420_50x35"]
want += [k[5:] for k in Z.files if k.startswith("file_pngx_") and ("adam7" in k or "_d16_" in k or "_c3_" in k)][:10]
for i, name in enumerate(want):
    open(f"{sys.argv[2]}/img{i}.bin", "wb").write(Z["file_" + name].tobytes())
print(len(want))
EOP
for f in $W/img*.bin; do $W/fuzz $f ${1:-3000}; done
