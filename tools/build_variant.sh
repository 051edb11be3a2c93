#!/bin/bash
# Diagnostic: build an A/B copy of the library with extra -D flags into build/variants/libksh_<name>.so (travels to the GPU box;
# select it with SHOGIDRL_AMD_LIB=...).  usage: tools/build_variant.sh <name> [-DKSH_VAR_...]...
set -e
R=$(cd $(dirname $0)/.. && pwd)
mkdir -p $R/build/variants
name=$1; shift
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC "$@" -o $R/build/variants/libksh_$name.so $R/shogidrl_amd/csrc/ksh_lib.hip
echo built $R/build/variants/libksh_$name.so
