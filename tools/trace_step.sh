set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export GPU_MAX_HW_QUEUES=6 WESUP_TRACE_BATCH=${1:-1}
rm -rf gpurun_out/trace_b${1:-1}
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_b${1:-1} -- python3 tools/step_trace.py run multi > gpurun_out/trace_b${1:-1}.log 2>&1 || { tail -5 gpurun_out/trace_b${1:-1}.log; exit 1; }
python3 tools/step_trace.py report gpurun_out/trace_b${1:-1} > gpurun_out/step_trace_b${1:-1}.txt
rm -rf gpurun_out/trace_b${1:-1}
tail -45 gpurun_out/step_trace_b${1:-1}.txt
