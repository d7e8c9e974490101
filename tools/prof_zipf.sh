cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_zipf -o zipf -- python3 tools/zipf_dump.py chain=2 diag=0 > gpurun_out/prof_zipf.log 2>&1
find gpurun_out/prof_zipf -name "*kernel_stats*" | head
