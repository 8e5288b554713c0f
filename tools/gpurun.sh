#!/bin/bash
# gpurun with the state of the tree stamped into .gitsha first (the GPU box gets no .git): profile summaries and bench
# records name the commit they were taken on.  usage: tools/gpurun.sh [--timeout S] -- '<command>'
cd "$(dirname "$0")/.."
sha=$(git rev-parse --short=12 HEAD)
git diff --quiet HEAD -- . ':!.gitsha' || sha="$sha-dirty"
echo "$sha" > .gitsha
exec /usr/local/graft/bin/gpurun "$@"
