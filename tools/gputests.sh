#!/bin/bash
# the -m gpu suite with its summary kept (developer tool): tools/gputests.sh [pytest args]
mkdir -p gpurun_out
python -m pytest tests -q -m gpu -p no:cacheprovider "$@" > gpurun_out/gputests.log 2>&1
echo "pytest rc=$?"
grep -E "passed|failed|error" gpurun_out/gputests.log | tail -5
grep -E "^FAILED|^ERROR" gpurun_out/gputests.log | head -20
