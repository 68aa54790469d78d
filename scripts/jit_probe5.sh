#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
bash scripts/profile_round.sh r02g 2>&1 | tail -3
