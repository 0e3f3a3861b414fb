# 8 x 32 against 16 x 16 patches on the layers whose width picks 16 x 16 (199-, 112-wide), flat tiles off
set -x
O=gpurun_out/r05z; mkdir -p $O
export L3_DEBUG_KNOBS=1
for pw in 16 32; do
L3_HALO_FLAT=0 L3_HALO_PW=$pw timeout 300 scripts/probes/halo_bench layers 128 > $O/layers_2d_pw$pw.txt 2>&1
done
timeout 300 scripts/probes/halo_bench layers 128 > $O/layers_default.txt 2>&1
paste -d'|' $O/layers_2d_pw16.txt $O/layers_2d_pw32.txt $O/layers_default.txt | cut -c1-34,35-52,77-92,134-152,177-192,234-252,277-292
