#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 600 python tools/dev/voter_threshold_probe.py 2>&1 | grep -v amdgpu.ids
