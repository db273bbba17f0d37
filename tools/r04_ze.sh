#!/bin/bash
set -u
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
( PFRL_TREE_SAMPLE=prefetch timeout 300 python $R/tools/per_dbg2.py ) 2>&1 | grep "per draw" | tail -n 2
( PER_DBG_LOAD=1 PFRL_TREE_SAMPLE=prefetch timeout 300 python $R/tools/per_dbg2.py ) 2>&1 | grep "per draw" | tail -n 2
( PER_DBG_LOAD=1 PFRL_TREE_SAMPLE=lean timeout 300 python $R/tools/per_dbg2.py ) 2>&1 | grep "per draw" | tail -n 2
