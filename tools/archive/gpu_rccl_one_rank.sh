#!/bin/bash
# The sharded field step on a one-rank RCCL process group, by itself on a fresh box.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
OPTILAND_TEST_RCCL=1 timeout 500 python -m pytest tests/test_gpu_rccl_one_rank.py -m gpu -q 2>&1 | tail -5
