#!/bin/bash
# A product library with different compile-time constants in some kernels, for same-box A/Bs (MCVD_LIB_PATH picks the library):
#   tools/build_variant.sh <tag> "<extra hipcc flags>" file1.cpp file2.cpp ...     -> mcvd_pytorch_amd/libmcvd_hip_<tag>.so
# (run after csrc/build.py: every other object comes from csrc/build/)
set -e
cd "$(dirname "$0")/../mcvd_pytorch_amd/csrc"
tag=$1; flags=$2; shift 2
mkdir -p build_$tag
skip=""
for f in "$@"; do
  b=$(basename $f .cpp)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I ../../include $flags -c kernels/$b.cpp -o build_$tag/$b.o &
  skip="$skip|$b.o"
done
wait
objs=$(ls build/*.o | grep -v -E "(${skip#|})$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmcvd_hip_$tag.so $objs build_$tag/*.o -ldl
ls -la ../libmcvd_hip_$tag.so
