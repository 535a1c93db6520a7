#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace stats + PMC HBM-traffic passes of bench.py.
# usage: tools/profile_round.sh <tag> [bench args...]   -> gpurun_out/<tag>/   (then tools/summarize_profile.py <tag>)
# Pass --warmup 0 so that the launches in the trace are exactly the launches of the one timed pass.
set -u
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py $* --no-cpu --no-pmc --no-warm --min-seconds 0 --max-repeats 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- $B > $out/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -o f -- $B --no-roofline > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -o w -- $B --no-roofline > $out/pmc_write.log 2>&1
find $out -name "*.csv" | head -20
grep -h -o '{"metric.*' $out/trace.log | cut -c1-300
