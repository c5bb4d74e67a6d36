#!/bin/bash
# run a command against two builds of the library on one box: ab/base.so, then ab/new.so (scratch files)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for v in base new base new; do cp ab/$v.so cuvs_amd/libcuvs_c.so; echo "== $v"; bash -c "$1"; done
