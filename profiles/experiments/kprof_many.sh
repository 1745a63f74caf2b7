#!/bin/bash
# per-kernel stats of several library builds on ONE box:  bash profiles/experiments/kprof_many.sh <tag> "<grep pattern>" "<bench args>" lib1 lib2 ...   ("-" = the in-tree library)
tag=$1; pat=$2; args=$3; shift 3
for lib in "$@"; do
  name=$(basename $lib .so); [ "$lib" = "-" ] && name=tree
  bash profiles/experiments/kprof.sh $tag/$name $lib "$pat" $args || exit 1
done
