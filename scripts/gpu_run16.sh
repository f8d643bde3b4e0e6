#!/bin/bash
for mb in 32 16 8 4; do
echo "REDGPU_HOST_CHUNK_MB=$mb"
REDGPU_HOST_CHUNK_MB=$mb python3 scripts/bench_host_path.py 2>&1 | grep -v amdgpu | grep "1 thread"
done
