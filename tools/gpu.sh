#!/bin/bash
# Local wrapper around gpurun (runs HERE, in the build container): rebuilds the in-tree libraries if stale and leaves
# the identity of what is about to travel in profiles/BUILD_ID -- .git stays behind, so the scripts that write
# summaries on the GPU box (tools/profile_round.sh, l2_round.sh, ...) read the commit and the library stamp from there.
# usage: tools/gpu.sh [--timeout S] -- '<command>'
cd "$(dirname "$0")/.."
python3 primitive3d_amd/_build.py > /dev/null || exit 1
dirty=""; [ -n "$(git status --porcelain -- primitive3d_amd include bench.py tools/*.sh tools/*.py 2>/dev/null)" ] && dirty="+dirty"
echo "$(git rev-parse HEAD)$dirty lib=$(cut -c1-16 primitive3d_amd/libp3dmc.so.stamp) $(date -u +%Y-%m-%dT%H:%MZ)" > profiles/BUILD_ID
exec /usr/local/graft/bin/gpurun "$@"
