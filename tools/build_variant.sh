#!/bin/bash
# build a kernel-library variant into build_var/<name>.so with extra -D flags:  tools/build_variant.sh name -DFOO=1 ...
# (build_var/ is git-ignored but travels to the GPU box; select a variant with CSTP_LIB_PATH=build_var/<name>.so)
set -e
name=$1; shift
mkdir -p build_var
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -shared -Iinclude -Icstp_amd/csrc "$@" -o build_var/$name.so \
  cstp_amd/csrc/igemm.hip cstp_amd/csrc/bn.hip cstp_amd/csrc/misc.hip cstp_amd/csrc/clip.hip cstp_amd/csrc/b16.hip
echo built build_var/$name.so
