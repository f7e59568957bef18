cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
python tools/probe/dbg_parity.py 2>&1 | tail -15
