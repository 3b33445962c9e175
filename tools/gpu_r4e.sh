mkdir -p gpurun_out/r4e
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(TCP|TA|TD|TCC|SQ|GRBM)_[A-Z0-9_]+" | sort -u > $GRAFT_REPO_ROOT/gpurun_out/r4e/counters.txt
wc -l $GRAFT_REPO_ROOT/gpurun_out/r4e/counters.txt
grep -E "^TA_|^TCP_|^TD_" $GRAFT_REPO_ROOT/gpurun_out/r4e/counters.txt | tr '\n' ' '
