#!/bin/bash
# Build the library locally (the in-tree .so travels with the snapshot), then run a command on the
# GPU box:  tools/gpu.sh [--timeout S] -- '<command>'
set -e
cd "$(dirname "$0")/.."
make -C gaia_seg_amd/csrc -j8 2>&1 | grep -E "error|Error" && { echo "BUILD FAILED"; exit 1; }
python -c "from gaia_seg_amd.hip import lib; lib.load()"
exec /usr/local/graft/bin/gpurun "$@"
