#!/bin/bash
# End-to-end batch of 1000 and 10 000 motifs (tools/e2e_profile.py), TXQ_TRACE for the 10 000 one: where the time goes.
cd "$GRAFT_REPO_ROOT" || exit 1
REPS=6 timeout -k 10 200 python tools/e2e_profile.py 1000 2>&1 | grep "^rep" | tail -3
REPS=5 TXQ_TRACE=1 timeout -k 10 300 python tools/e2e_profile.py 10000 2>&1 | grep "^rep\|^\[txq\]" | tail -8
