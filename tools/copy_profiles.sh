# gpurun_out/prof_<tag>/ (what tools/collect_profiles.sh <tag> left on the GPU box) -> profiles/<tag>_*:  bash tools/copy_profiles.sh r05 [r05_c4 ...]
for R in "$@"; do
  O=gpurun_out/prof_$R
  [ -d $O ] || { echo "no $O"; continue; }
  cp $O/kernel_stats.csv profiles/${R}_bench_kernel_stats.csv
  cp $O/bench_line.json profiles/${R}_bench_line.json
  cp $O/bench_under_rocprof.json profiles/${R}_bench_under_rocprof.json
  cp $O/${R}_roofline_inputs.json profiles/${R}_roofline_inputs.json
  cp $O/step_traffic.txt profiles/${R}_step_traffic.txt
  cp $O/wino_table.txt profiles/${R}_wino_table.txt
  [ -f $O/${R}_layer_mfma.csv ] && cp $O/${R}_layer_mfma.csv profiles/${R}_layer_mfma.csv
  [ -f $O/layer_table.txt ] && cp $O/layer_table.txt profiles/${R}_layer_table_direct.txt
  echo "$R copied"
done
