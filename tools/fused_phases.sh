# Phases of the one-kernel Winograd product route per block (tools/probes/fused_phases.hip): gpurun -- 'bash tools/fused_phases.sh'
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/fused_phases; mkdir -p $O
F="-O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-result -Wno-unused-value -I wesup_amd/csrc -I include"
/opt/rocm/bin/hipcc $F -c tools/probes/fused_phases.hip -o $O/fp.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 $O/fp.o wesup_amd/csrc/plan.o -o $O/fused_phases || exit 1
for shape in "64 64 480 4" "64 128 240 4" "128 128 240 4" "256 256 120 4" "64 64 1024 8"; do timeout -k 10 60 $O/fused_phases $shape || exit 1; done | tee $O/phases.txt
rm -f $O/fp.o $O/fused_phases
