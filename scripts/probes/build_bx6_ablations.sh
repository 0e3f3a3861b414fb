# timing-only ablations of conv_wino_bx6.hip: libl3hip_bx6_<abl>.so under scripts/probes/ (L3_DEBUG_KNOBS=1 L3_LIB_PATH=... selects one)
set -e
cd "$(dirname "$0")/../.."
OBJ=l3embedding_amd/lib/obj
for A in ${ABLS:-NOVALU NOLDS NODMA NOBAR NOMFMA} ; do
  defs=$(echo $A | sed 's/+/ -DBX6_ABL_/g')
  tag=$(echo $A | tr '+' '_')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DBX6_ABL_$defs -c l3embedding_amd/csrc/conv_wino_bx6.hip -o /tmp/bx6_$tag.o
  objs=$(ls $OBJ/*.o | grep -v conv_wino_bx6.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/probes/libl3hip_bx6_$tag.so $objs /tmp/bx6_$tag.o -ldl
  echo built $tag
done
