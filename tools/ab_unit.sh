#!/bin/bash
# Same-box A/B of ONE translation unit: tools/ab_unit.sh <unit.hip> [batch] [rounds]
# _ab_old/<unit.hip> (git-ignored; e.g. `git show HEAD:healnet_amd/csrc/<unit.hip> > _ab_old/<unit.hip>`) is compiled on the GPU box against
# the current headers and linked with the other current objects into /tmp/libhn_old.so; tools/bench_chain.py (latent block + cfg2
# forward, HIP-event timed) then alternates between the two libraries through HN_LIB_PATH.
U=$1; B=${2:-32}; N=${3:-4}; R=${GRAFT_REPO_ROOT:-$(pwd)}
python -c "from healnet_amd import _capi; _capi.build(force=True)" 2>&1 | tail -1
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form=1"
/opt/rocm/bin/hipcc $F -I $R/healnet_amd/csrc -I $R/include -c $R/_ab_old/$U -o /tmp/old_unit.o 2>/dev/null || { echo "old unit failed to compile"; exit 1; }
OBJS=$(ls $R/healnet_amd/build/*.hip.o | grep -v "/$U.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libhn_old.so $OBJS /tmp/old_unit.o || exit 1
for i in $(seq $N); do
  HN_LIB_PATH=/tmp/libhn_old.so python tools/bench_chain.py $B 2>/dev/null | tail -1
  python tools/bench_chain.py $B 2>/dev/null | tail -1
done
