#!/bin/bash
# Same-box A/B of ONE translation unit: tools/ab_unit.sh <unit.hip> [batch] [rounds] [more variants ...]
# _ab_old/<unit.hip> (git-ignored; e.g. `git show HEAD:healnet_amd/csrc/<unit.hip> > _ab_old/<unit.hip>`) -- and every further file named
# after the rounds, also under _ab_old/ -- is compiled on the GPU box against the current headers and linked with the other current
# objects into /tmp/libhn_<name>.so; tools/bench_chain.py (latent block + cfg2 forward, HIP-event timed) then alternates between those
# libraries and the in-tree one through HN_LIB_PATH.
U=$1; B=${2:-32}; N=${3:-4}; shift 3 2>/dev/null; R=${GRAFT_REPO_ROOT:-$(pwd)}
python -c "from healnet_amd import _capi; _capi.build(force=True)" 2>&1 | tail -1
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form=1"
OBJS=$(ls $R/healnet_amd/build/*.hip.o | grep -v "/$U.o")
LIBS=""
for V in $U "$@"; do
  cp $R/_ab_old/$V /tmp/ab_$V.hip
  /opt/rocm/bin/hipcc $F -I $R/healnet_amd/csrc -I $R/include -c /tmp/ab_$V.hip -o /tmp/ab_$V.o 2>/dev/null || { echo "variant $V failed to compile"; exit 1; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libhn_$V.so $OBJS /tmp/ab_$V.o || exit 1
  LIBS="$LIBS /tmp/libhn_$V.so"
done
for i in $(seq $N); do
  for L in $LIBS; do HN_LIB_PATH=$L python tools/bench_chain.py $B 2>/dev/null | tail -1; done
  python tools/bench_chain.py $B 2>/dev/null | tail -1
done
