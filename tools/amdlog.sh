cd "$GRAFT_REPO_ROOT"
export WESUP_SHALLOW_G_AT=9 WESUP_SIDE_BLOCK_AT=1 WESUP_TRACE_BATCH=4
AMD_LOG_LEVEL=4 timeout -k 10 200 python3 tools/step_trace.py run multi > /tmp/amdlog.txt 2>&1
wc -l /tmp/amdlog.txt
# the last step: lines after the last-but-one sgd
grep -n "sgd_kernel" /tmp/amdlog.txt | tail -3
L=$(grep -n "ShaderName : .*sgd_kernel" /tmp/amdlog.txt | tail -2 | head -1 | cut -d: -f1)
tail -n +$L /tmp/amdlog.txt | grep -n "ShaderName\|Barrier\|barrier\|Marker\|marker\|signal\|Signal" | cut -c1-230 > gpurun_out/amdlog_step.txt
wc -l gpurun_out/amdlog_step.txt
grep -n "dual_transform" gpurun_out/amdlog_step.txt | head -3
