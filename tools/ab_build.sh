#!/bin/bash
# A/B build of libisls_hip.so with extra compiler flags into ab/<name>/ (git-ignored; select with ISLS_HIP_LIB):
#     tools/ab_build.sh diag -DISLS_DIAG                       everything rebuilt with the flags
#     ISLS_AB_ONLY="riccati" tools/ab_build.sh g1 -DFOO=1      only riccati.hip rebuilt, the other objects taken from the in-tree build
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/ab/$name
mkdir -p "$out"
cd "$root/ilqr-admm_amd/csrc"
FL="-O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wno-unused-variable -Wno-unused-but-set-variable -Wno-unused-value $*"
SRCS="riccati riccati_ff riccati_ffrec ff_segments rollout admm project sls_admm feedback_columns misc generic capi"
FAMS="4_2_0 4_2_2 4_2_3 4_2_4 9_3_0 9_3_1 6_3_0 6_3_3 2_1_0 2_1_3 3_1_0 6_2_0 2_2_0 3_3_0"
pids=()
throttle() { if (( ${#pids[@]} >= 5 )); then wait "${pids[0]}"; pids=("${pids[@]:1}"); fi; }
for s in $SRCS; do
  if [[ -n "$ISLS_AB_ONLY" && " $ISLS_AB_ONLY " != *" $s "* ]]; then cp $s.o "$out/$s.o"; continue; fi
  /opt/rocm/bin/hipcc $FL -c $s.hip -o "$out/$s.o" & pids+=($!); throttle
done
for f in $FAMS; do
  if [[ -n "$ISLS_AB_ONLY" && " $ISLS_AB_ONLY " != *" rollout_f "* ]]; then cp rollout_f_$f.o "$out/rollout_f_$f.o"; continue; fi
  IFS=_ read nx nu md <<< "$f"
  /opt/rocm/bin/hipcc $FL -DISLS_FAM_NX=$nx -DISLS_FAM_NU=$nu -DISLS_FAM_MODEL=$md -c rollout_family.hip -o "$out/rollout_f_$f.o" & pids+=($!); throttle
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/libisls_hip.so" "$out"/*.o
echo "$out/libisls_hip.so"
