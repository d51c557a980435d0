#!/bin/bash
# builds the timing library and runs tools/xch_phases.py
cd "$GRAFT_REPO_ROOT"
mkdir -p abtest
id=$(python -c "import __graft_entry__ as g; print(g.source_hash())")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DGMPNP_XTIMING -DGMPNP_DEV_HOOKS -DGMPNP_BUILD_ID="\"$id\"" -Iinclude -Igmpnp_amd/csrc gmpnp_amd/csrc/gmpnp_api.hip gmpnp_amd/csrc/gmpnp_topology.cpp -ldl -o abtest/lib_xt.so || exit 1
GMPNP_LIB=$PWD/abtest/lib_xt.so timeout -k 10 200 python tools/xch_phases.py 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo"
