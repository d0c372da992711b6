for rep in 1 2; do
for v in v_prev v_ct_relu_newmask c1_relu4 c2_med3_pinned; do
  WESUP_HIP_LIB=$PWD/wesup_amd/csrc/variants/$v.so timeout -k 10 120 python tools/layer_table.py 10 2>&1 | tail -1 | sed "s/^/$v /"
done; done
