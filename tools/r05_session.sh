#!/bin/bash
# Round-5 GPU session: the -m gpu suite, the headline bench (no extras / cpu baseline: those run at round end), one training
# iteration as a timeline.  Logs under gpurun_out/.  Usage: gpurun -- 'bash tools/r05_session.sh [tag] [what...]'
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-s}; shift
WHAT=${@:-"tests bench timeline"}
mkdir -p $O
cd $R
for w in $WHAT; do
  case $w in
    tests)    timeout 900 python -m pytest tests -m gpu -q -x --no-header -p no:cacheprovider > $O/${TAG}_tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/${TAG}_tests.log ;;
    testsall) timeout 1200 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider > $O/${TAG}_tests.log 2>&1; echo "tests rc=$?"; tail -15 $O/${TAG}_tests.log ;;
    bench)    timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/${TAG}_bench.log 2>&1; echo "bench rc=$?"; tail -1 $O/${TAG}_bench.log | cut -c1-400 ;;
    benchfull) timeout 900 python bench.py --steps 20 --warmup 5 > $O/${TAG}_benchfull.log 2>&1; echo "benchfull rc=$?"; tail -1 $O/${TAG}_benchfull.log | cut -c1-300 ;;
    timeline) timeout 600 bash tools/prof_timeline.sh; cp $O/timeline.csv $O/${TAG}_timeline.csv; cp $O/timeline_summary.txt $O/${TAG}_timeline_summary.txt ;;
    *)        echo "custom: $w"; timeout 900 bash -c "$w" > $O/${TAG}_custom.log 2>&1; tail -20 $O/${TAG}_custom.log ;;
  esac
done
