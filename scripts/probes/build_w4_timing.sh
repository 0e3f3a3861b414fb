# Builds instrumented copies of libl3hip.so (scripts/probes/w4_instrument.py) for scripts/probes/w4_timing.py:
#   bash scripts/probes/build_w4_timing.sh "<abl>:<wave> ..."     e.g. ":0 :8 nostore:0 noprefetch:0"
set -e
R=$(cd $(dirname $0)/../.. && pwd)
T=/tmp/w4_timing_build; mkdir -p $T; cp $R/l3embedding_amd/csrc/*.h $T/
OBJS=$(ls $R/l3embedding_amd/lib/obj/*.o | grep -v conv_wino4.o)
for v in ${1:-":0 :8"}; do
  abl=${v%%:*}; wave=${v##*:}
  W4_ABL=$abl python $R/scripts/probes/w4_instrument.py $T/w4_${abl}_$wave.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DW4_TWAVE=$wave -I$R/include -c $T/w4_${abl}_$wave.hip -o $T/w4_${abl}_$wave.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/scripts/probes/libl3hip_timing_${abl:-asbuilt}_w$wave.so $OBJS $T/w4_${abl}_$wave.o -ldl
  echo built libl3hip_timing_${abl:-asbuilt}_w$wave.so
done
