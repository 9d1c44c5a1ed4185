#!/bin/bash
# developer tool: tools/trace.sh with an alternative library:  gpurun -- tools/trace_expt.sh <name> <bench args>   (tools/build_expt.sh <name> ... first)
export GPUDRIVE_DEV=1 GPUDRIVE_AMD_LIB=$GRAFT_REPO_ROOT/build/expt/expt_$1.so
[ -f "$GPUDRIVE_AMD_LIB" ] || { echo "build it first: tools/build_expt.sh $1 ..."; exit 1; }
shift
exec bash "$(dirname "$0")/trace.sh" "$@"
