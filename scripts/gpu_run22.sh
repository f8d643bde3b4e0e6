#!/bin/bash
set -u
bash scripts/gpu_run12.sh || exit 1
bash scripts/profile_r02.sh 3 log100 20 || exit 1
bash scripts/profile_lists.sh || exit 1
bash scripts/secondary_benchmarks.sh
