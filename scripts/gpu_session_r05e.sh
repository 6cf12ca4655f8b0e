#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05e; mkdir -p $O
timeout 300 python scripts/probes/ns1_grad_diag.py 2>&1 | grep -v "amdgpu.ids\|Warning\|detach\|errs =" | tee $O/ns1_diag.log | cut -c1-330
