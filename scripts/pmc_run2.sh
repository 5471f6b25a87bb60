#!/bin/bash
# scripts/pmc_run2.sh "<python script>" <kernel filter> "<counter group 1>" ["<group 2>" ...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
s=$1; f=$2; shift 2
i=0
for grp in "$@"; do
  i=$((i+1)); rm -rf /tmp/pq$i
  rocprofv3 --pmc $grp -d /tmp/pq$i -o k --output-format csv -- python $R/$s > /tmp/pq$i.log 2>&1
  tail -1 /tmp/pq$i.log
  python $R/scripts/pmc_summary.py /tmp/pq$i 2>&1 | grep -A8 "$f"
done
