#!/bin/bash
# tools/gpurun.sh [--timeout S] -- <command>: gpurun, with the commit this tree stands at written to gpurun_commit.txt first (the GPU box
# gets a snapshot without .git; tools/pmc_traffic.py and bench.py quote it as the commit a profile was taken at)
cd "$(dirname "$0")/.."
{ git rev-parse --short=12 HEAD; git diff --quiet HEAD -- . ':!gpurun_commit.txt' || echo "+uncommitted"; } | tr '\n' ' ' | sed 's/ $//' > gpurun_commit.txt
exec /usr/local/graft/bin/gpurun "$@"
