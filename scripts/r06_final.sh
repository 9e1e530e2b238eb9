#!/bin/bash
# round 6, last sources: full GPU suite (with the parity printouts), rocprofv3 stats + PMC traffic of the bench command, the bench line twice, the 1024^2 ROI bench
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06z; rm -rf "$O"; mkdir -p "$O"
timeout 2400 python3 -m pytest tests -m gpu -q -s -p no:cacheprovider > "$O/gputest_full.log" 2>&1; echo "pytest rc $?" | tee -a "$O/gputest_full.log"
tail -3 "$O/gputest_full.log"
bash scripts/final_profiles.sh > "$O/final_profiles.log" 2>&1
python3 bench.py > "$O/bench_run1.json" 2> "$O/bench_run1.err"; tail -c 600 "$O/bench_run1.json"
python3 bench.py > "$O/bench_run2.json" 2> "$O/bench_run2.err"
timeout 600 python3 scripts/bench_roi1024.py > "$O/roi1024.txt" 2>&1; tail -3 "$O/roi1024.txt"
