#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r03m; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "lockstep or grid_sync or captured" > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt
tail -15 $O/pytest.txt
