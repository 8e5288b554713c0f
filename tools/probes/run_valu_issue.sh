#!/bin/bash
# builds (if needed) and runs the issue-rate probe on the GPU box; output -> gpurun_out/valu_issue.txt
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/probes/valu_issue.hip -o /tmp/valu_issue && /tmp/valu_issue | tee gpurun_out/valu_issue.txt
