#!/bin/bash
timeout 300 python scripts/ways_priority_probe.py 2>&1 | grep -v amdgpu
