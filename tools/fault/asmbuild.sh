#!/bin/bash
# asmbuild.sh NAME FILE.s : links a variant library whose device code of cvec_decode.hip comes from FILE.s
set -e
name=$1; src=$2
LLVM=/opt/rocm/lib/llvm/bin
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wall -Wno-unused-function"
OTHERS="cvec_api.o cvec_vstream.o cvec_tailfix.o cvec_encode.o cvec_filter_ops.o cvec_stream.o cvec_appender.o"
$LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $src -o var/dev_$name.o
$LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o var/dev_$name.out var/dev_$name.o
$LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=var/dev_$name.out -output=var/dev_$name.hipfb
/opt/rocm/bin/hipcc $F --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang var/dev_$name.hipfb -c -o var/decode_$name.o cvec_decode.hip
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../lib/libcvec_$name.so var/decode_$name.o $OTHERS
echo built $name
