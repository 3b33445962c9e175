#!/bin/bash
# gpurun with the tree's identity on board: the GPU box gets a snapshot WITHOUT .git, so the commit the measurements belong to is
# written to .git_head (git-ignored, travels with the snapshot) right before the call; bench.py / tools/pmc_bench.sh stamp it into
# what they write ("at commit ?").   usage: tools/grun.sh [--timeout S] -- '<command>'
cd "$(dirname "$0")/.." || exit 1
H=$(git rev-parse --short=12 HEAD 2>/dev/null || echo unknown)
git diff --quiet HEAD 2>/dev/null || H="$H+dirty($(git diff HEAD | sha256sum | cut -c1-8))"
echo "$H" > .git_head
exec /usr/local/graft/bin/gpurun "$@"
