#!/bin/bash
# gfx950 disassembly of a hipcc object:  tools/devasm.sh path/to/x.o > x.s   (then tools/isa_profile.py x.s <mangled name part>)
set -e
O=$(readlink -f "$1"); D=$(mktemp -d); cp "$O" $D/x.o
(cd $D && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading x.o > /dev/null)
/opt/rocm/lib/llvm/bin/llvm-objdump -d $D/x.o.0.hipv4-amdgcn-amd-amdhsa--gfx950
rm -rf $D
