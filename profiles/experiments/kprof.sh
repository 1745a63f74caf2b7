#!/bin/bash
# per-kernel stats of one library build:  bash profiles/experiments/kprof.sh <tag> <lib.so|-> <grep pattern> [bench args]
tag=$1; lib=$2; pat=$3; shift 3
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
[ "$lib" != "-" ] && export CVAE_LIB=$lib
rocprofv3 --kernel-trace --stats --output-format csv -d $out/p -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-probe --no-fwd-bwd-rate "$@" > $out/b.log 2>&1 || { tail -5 $out/b.log; exit 1; }
python3 profiles/kstats.py $out/p > $out/k.txt
echo "== $tag"; grep -E "$pat" $out/k.txt
