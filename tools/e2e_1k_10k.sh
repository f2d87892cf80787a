#!/bin/bash
# The end-to-end k = 4 batch at 1000 and 10000 motifs (best five of ten / best three of five runs, ms), and the host's
# expansion alone over threads (tools/host_scaling.sh).
cd "$GRAFT_REPO_ROOT" || exit 1
REPS=10 timeout -k 10 120 python3 tools/e2e_profile.py 2>/dev/null | grep "^rep" | awk '{print $3}' | sort -n | head -5 | tr '\n' ' '; echo
REPS=5 timeout -k 10 120 python3 tools/e2e_profile.py 10000 2>/dev/null | grep "^rep" | sort -n -k3 | head -3 | cut -c1-260
