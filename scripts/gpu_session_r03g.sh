#!/bin/bash
timeout 300 python scripts/fps_interference_probe.py 2>&1 | grep -v amdgpu
timeout 300 python scripts/fps_interference_probe.py 2>&1 | grep -v amdgpu
