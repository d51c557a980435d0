#!/bin/bash
# A/B builds of libgmpnp.so for same-box comparisons: tools/ab_build.sh <name> [<git-rev>|WORK] [-D flags...]
# -> abtest/lib_<name>.so, stamped with the CURRENT tree's build id so that bench.py (GMPNP_LIB=...) accepts it.
set -e
cd "$(dirname "$0")/.."
name=$1; rev=${2:-WORK}; shift; shift || true
id=$(python -c "import __graft_entry__ as g; print(g.source_hash())")
src=gmpnp_amd/csrc; inc=include
if [ "$rev" != "WORK" ]; then
  tmp=$(mktemp -d); git archive "$rev" gmpnp_amd/csrc include | tar -x -C "$tmp"; src=$tmp/gmpnp_amd/csrc; inc=$tmp/include
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DGMPNP_BUILD_ID="\"$id\"" "$@" -I$inc -I$src $src/gmpnp_api.hip $src/gmpnp_topology.cpp -ldl -o abtest/lib_$name.so
echo built abtest/lib_$name.so
