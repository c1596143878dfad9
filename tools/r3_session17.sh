#!/bin/bash
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s17
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -25 $OUT/pytest.log | cut -c1-200
bash $R/tools/r3_quick.sh "adam|FcWgradOp<2, 2, 1, 2, 5>|reduce_parts|fc_epilogue|finalize"
