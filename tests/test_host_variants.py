"""The host's staged expansion under settings that small tests would otherwise never reach: the
process-wide knobs are read once, so the parity suites are run again in a child process with them set."""
import os
import subprocess
import sys

from conftest import ROOT


def _rerun(env_extra, *files):
    env = dict(os.environ, **env_extra)
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", *[os.path.join(ROOT, "tests", f) for f in files]]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-1000:]


def test_parity_suites_with_the_no_merge_decision_on_tiny_lists():
    """TETREX_MERGE_SAMPLE=4: every join list of 4+ states is asked whether merging pays, so receivers
    without a merge table (QueryExpansion::merging_pays, `append_only`) occur all over the parity tests."""
    _rerun({"TETREX_MERGE_SAMPLE": "4"}, "test_host_staged.py", "test_host_gaps.py", "test_fuzz_parity.py")


def test_parity_suites_without_overlap_and_with_forced_verified_levels():
    _rerun({"TETREX_NO_OVERLAP": "1", "TETREX_VERIFIED_LEVELS": "1", "TETREX_THREADS": "3"}, "test_host_staged.py", "test_host_gaps.py")
