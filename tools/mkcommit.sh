#!/bin/bash
# builds tools/abl_<name>.so from the kernel sources of a git commit (same-box A/B against an earlier state: tools/ab.sh)
# usage: tools/mkcommit.sh <commit> <name>
set -e
R=$(cd $(dirname $0)/.. && pwd)
T=$(mktemp -d /tmp/ccsp_commit_XXXX)
mkdir -p $T/csrc $T/include
for f in $(git -C $R ls-tree --name-only $1 diffusion-ccsp_amd/csrc/ | grep -E "\.(h|hip)$"); do git -C $R show $1:$f > $T/csrc/$(basename $f); done
git -C $R show $1:include/ccsp.h > $T/include/ccsp.h
sed -i "s#\"../../include/ccsp.h\"#\"$T/include/ccsp.h\"#" $T/csrc/ccsp_hip.hip
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -pthread -o $R/tools/abl_$2.so $T/csrc/ccsp_hip.hip
rm -rf $T
echo built $R/tools/abl_$2.so
