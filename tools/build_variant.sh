#!/bin/bash
# Build a variant of the library with extra -D flags on ONE source file, for same-box A/B runs (tools/ab_libs.sh):
#   tools/build_variant.sh <name> <source.hip> "<flags>"   ->   ab_libs/lib_<name>.so   (git-ignored; travels with gpurun)
set -e
R=$(cd $(dirname $0)/.. && pwd)
S=$R/crowdmod-ddpm-4d_amd/csrc
mkdir -p $R/ab_libs
name=$1; src=$2; flags=$3
obj=$R/ab_libs/${name}_$(basename $src .hip).o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -Wno-unused-value -DCM_PD27=3 -DCM_PD8=2 $flags -c $S/$src -o $obj
others=$(ls $S/*.o | grep -v "/$(basename $src .hip).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $others $obj -o $R/ab_libs/lib_$name.so
echo built ab_libs/lib_$name.so
