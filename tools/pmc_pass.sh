#!/bin/bash
# usage: tools/pmc_pass.sh <tag> <counters...> -- <python args>   (one rocprofv3 --pmc pass, csv under gpurun_out/pmc_<tag>)
tag=$1; shift
ctrs=()
while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc "${ctrs[@]}" --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag" -- python3 "$@"
