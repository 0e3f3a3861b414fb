# Builds scripts/probes/libl3hip_stamps[_ablN].so (the product objects with conv_bf16_halo.hip recompiled -DL3_HALO_STAMPS: wave 0 of
# every halo block stamps the wall clock at its phase boundaries) and scripts/probes/halo_bench_stamps[_ablN] linked against it.
#   bash scripts/probes/build_halo_stamps.sh [abl ...]     abl = timing-only ablations of the 64-channel register-filter kernel
#                                                          (HALO_ABL bit 0: no filter loads in the tap loop, bit 1: no halo reads)
#   L3_DEBUG_KNOBS=1 HB_STAMPS=1 scripts/probes/halo_bench_stamps layers 128
set -e
R=$(cd $(dirname $0)/../.. && pwd)
T=/tmp/halo_stamps_build; mkdir -p $T
OBJS=$(ls $R/l3embedding_amd/lib/obj/*.o | grep -v conv_bf16_halo.o)
for abl in "" "$@"; do
  sfx=${abl:+_abl$abl}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DL3_HALO_STAMPS ${abl:+-DHALO_ABL=$abl} -I$R/include -c $R/l3embedding_amd/csrc/conv_bf16_halo.hip -o $T/conv_bf16_halo_stamps$sfx.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/scripts/probes/libl3hip_stamps$sfx.so $OBJS $T/conv_bf16_halo_stamps$sfx.o -ldl
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 $R/scripts/probes/halo_bench.cpp -I$R/l3embedding_amd/csrc -I$R/include -L$R/scripts/probes -ll3hip_stamps$sfx -ldl -Wl,-rpath,'$ORIGIN' -o $R/scripts/probes/halo_bench_stamps$sfx
  echo built halo_bench_stamps$sfx
done
