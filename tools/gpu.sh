#!/bin/bash
# gpurun wrapper: stale bytecode and caches never travel to the GPU box (.gpurunignore lists them too)
cd "$(dirname "$0")/.." || exit 1
find . -name __pycache__ -type d -prune -exec rm -rf {} + 2>/dev/null
rm -rf .pytest_cache .hypothesis dolfinx_materials_amd/_jit
exec /usr/local/graft/bin/gpurun "$@"
